#!/usr/bin/env python3
"""Race check for the LDS-DMA kernels: the same 16-bit SalUNet evaluation (64 AV clips, BASELINE configs[4] shapes) repeated back to back
with the chip loaded must give the same bits every time -- every kernel of the step sums in a fixed order, so a differing output is a
DMA ring overwritten while it was being read (that is how the three-slot ring of conv16_dma was found).  GPU only.
usage: tools/stress_repeat.py [--batch 64] [--reps 12] [--precision fp16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = sys.argv[1:]
    B, reps, prec = 64, 12, "fp16"
    i = 0
    while i < len(args):
        if args[i] == "--batch":
            B = int(args[i + 1]); i += 2
        elif args[i] == "--reps":
            reps = int(args[i + 1]); i += 2
        elif args[i] == "--precision":
            prec = args[i + 1]; i += 2
        else:
            i += 1
    dev = torch.device("cuda", 0)
    cfg = bench.Config()
    net, _ = bench.build_net(cfg, dev)
    net.compute_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[prec]
    g = torch.Generator(device="cpu").manual_seed(77)
    H, W = cfg.img_size
    x = torch.randn((B, 1, H, W), generator=g).to(dev)
    feats = [torch.randn((B, c, 8, H // s, W // s), generator=g).to(dev) for c, s in zip(cfg.up_channel, (32, 16, 8, 4))]
    audio = torch.randn((B, 512, 9, H // 32, W // 32), generator=g).to(dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.long)
    with torch.no_grad():
        ref = net(x, t, feats, audio).clone()
        bad = 0
        for r in range(reps):
            out = net(x, t, feats, audio)
            same = torch.equal(out, ref)
            bad += 0 if same else 1
            if not same:
                d = (out.float() - ref.float()).abs()
                print(f"rep {r}: DIFFERS in {int((d > 0).sum())} elements, max {d.max().item():.3e}", flush=True)
    torch.cuda.synchronize()
    print(f"{prec} B={B}: {reps} repeats, {bad} differing", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
