"""BASELINE.json configs at their FULL sizes (224x384, all four stages at reference widths).

configs[2]/[3] per-GPU shape (B=4) against the CPU oracle on one clip of the batch, and configs[4] (64 clips per GPU,
audio-visual, fp16 storage; evaluated in ONE pass of 64 clips, ``SalUNet.clips_per_pass``) through size-independent
properties: chunked == per-clip, determinism, range, bounded peak memory.
"""
import pytest
import torch

from oracle import salunet_oracle as orc
from tests.test_gpu_lowp import build as build_dt

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(B, av, tag):
    cfg = orc.SalUNetConfig()
    x, feats, audio = orc.synth_inputs(cfg, B, av, tag=tag)
    return cfg, x, feats, audio


def test_full_size_batch4_matches_oracle_on_one_clip():
    """configs[2] shape (AV, B=4, 224x384): clip 2 of the batch against the fp32 CPU oracle, 1e-3 relative (north_star)."""
    cfg, x, feats, audio = _inputs(4, True, "cfg2")
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    net = build_dt(cfg, sd, torch.float32)
    t = torch.tensor([999.0, 640.25, 17.0, 0.0])
    with torch.no_grad():
        out = net(x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], audio.to(DEV))
        ref = orc.salunet_forward(sd, cfg, x[2:3], t[2:3], [f[2:3] for f in feats], audio[2:3])
    err = (out[2:3].cpu() - ref).abs().max().item()
    print("full-size B=4 clip 2 vs oracle: max abs", err)
    assert err < 1e-3 * ref.abs().max().item()


@pytest.mark.parametrize("dname", ["fp32", "fp16", "bf16"])
def test_config4_batch64_av_full_size(dname):
    """configs[4]: 64 clips on one GPU, audio-visual, full size.  fp32 is the parity arithmetic, fp16 the configuration
    BASELINE names, bf16 its sibling."""
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dname]
    B = 64
    cfg = orc.SalUNetConfig()
    g = torch.Generator().manual_seed(64)      # 2 GB of inputs: a seeded torch stream (the closed-form fill is fp64-slow)
    x = torch.randn((B, 1, 224, 384), generator=g)
    feats = [torch.randn((B, c, 8, 224 // s, 384 // s), generator=g) for c, s in zip(cfg.up_channel, (32, 16, 8, 4))]
    audio = torch.randn((B, 512, 9, 7, 12), generator=g)
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    net = build_dt(cfg, sd, dt)
    # one pass of 64 on every datapath that takes the tap form (fp32); 16-bit storage uses the direct form: 2 bytes per element
    cpp = net.clips_per_pass(net._use_tap_conv(None, "mt"))
    print(f"config4 {dname}: clips per pass {cpp}")
    assert cpp == 64
    xd, fd, ad = x.to(DEV), [f.to(DEV) for f in feats], audio.to(DEV)
    del x, feats, audio
    t = (torch.arange(B, device=DEV) * 15) % 1000
    with torch.no_grad():
        net(xd[:16], t[:16], [f[:16] for f in fd], ad[:16])      # warm-up: weight packing, allocator pools
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        o64 = net(xd, t, fd, ad)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        o64b = net(xd, t, fd, ad)
        assert torch.equal(o64, o64b)                             # deterministic
        assert o64.shape == (B, 1, 224, 384) and o64.dtype == torch.float32
        assert torch.isfinite(o64).all() and o64.min() > 0 and o64.max() < 1
        # chunked evaluation == per-clip evaluation (clips are independent in eval mode)
        # fp32: only the summation order differs between a 1-clip and a 16-clip pass (tile / split-K plans depend on M).
        # 16-bit storage: the same order difference is rounded to the storage type at every layer, so the two
        # evaluations agree to the datapath's own end-to-end tolerance (tests/test_gpu_lowp.py::LOWP_ATOL), not bitwise.
        tol = {"fp32": 1e-5, "fp16": 4e-3, "bf16": 3e-2}[dname]
        for i in (0, 15, 16, 37, 63):
            oi = net(xd[i:i + 1], t[i:i + 1], [f[i:i + 1] for f in fd], ad[i:i + 1])
            d = (oi - o64[i:i + 1]).abs().max().item()
            assert d <= tol, (dname, i, d)
        # and different clips do give different maps
        assert (o64[0] - o64[1]).abs().max().item() > 1e-3
    # peak working set of the 64-clip pass: the tap products / the 16-bit 4-scale sum [64,112,192,768] (2.1 GB) + their
    # producers; sized for 288 GB of HBM: bound = 40 GB fp32, half for 16-bit storage
    bound = 40e9 if dname == "fp32" else 20e9
    print(f"config4 {dname}: peak extra memory {peak / 1e9:.2f} GB")
    assert peak < bound
