#!/usr/bin/env python3
"""Headline benchmark: denoise-steps/s of the DiffSal sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): visual-only, batch 4 clips per GPU, 224x384, 50-NFE
DPM-Solver trajectories (49 multistep-2 steps, logSNR grid, + denoise-to-zero, model_type x_start), fp32,
faithful full graph (all 9 frames every step, no cross-step caching), synthetic N(0,1) inputs of the
shapes in R/models/diff_model.py:105-111, seeded random-init weights of the reference architecture (no checkpoints offline).

One "step" = one denoising step of the sampler on one batch: one SalUNet evaluation + its solver update.
Trajectories run back to back; exactly K steps are timed (a trailing partial trajectory is cut after its
last needed evaluation).  value = n_gpus * batch * K / max-over-ranks(wall time), inputs resident in HBM.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (fp32-MFMA implicit GEMM): algorithmic FLOPs of the launches of one
                  timed step / their HIP-event durations, vs the 157.3 TFLOP/s dense fp32 matrix peak.
  cpu_baseline -- the CPU oracle (oracle/salunet_oracle.py, "port") timed on this box's host cores on a
                  bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, dense bf16 matrix peak
NFE_PER_TRAJECTORY = 50


class _Stop(Exception):
    pass


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# SURVEY 8(a) rows that are dense contractions (implicit-GEMM kernel); everything else is a streaming kernel
# "K10f" = block_front (K8 + K9 + K10 + K11 of a stage in one MFMA launch)
GEMM_CLASSES = ("K2", "K4", "K5", "K7-align", "K10", "K10f", "K12", "K13", "K14", "gemm", "wgrad")


def class_table(events, mfma_peak_tflops):
    """Per kernel class: launches, algorithmic FLOPs and once-through bytes (SURVEY 8d denominators) of ONE
    step, HIP-event time of exactly those launches, and the achieved fraction of BOTH rooflines; `bound` names the
    roofline the class sits closer to, `frac` its fraction there."""
    agg = {}
    for e0, e1, fl, cls, nb, *_ in events:
        a = agg.setdefault(cls, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += fl
        a[2] += nb
        a[3] += e0.elapsed_time(e1)
    rows = []
    for cls in sorted(agg, key=lambda c: -agg[c][3]):
        n, fl, nb, ms = agg[cls]
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        gbs = nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        f_m, f_h = tf / mfma_peak_tflops, gbs / HBM_PEAK_GBS
        bound = "mfma" if (cls in GEMM_CLASSES and f_m >= f_h) else "hbm"
        rows.append({"class": cls, "bound": bound, "launches": n, "gflop": round(fl / 1e9, 3), "mbytes": round(nb / 1e6, 2),
                     "ms": round(ms, 4), "tflops": round(tf, 2), "gbs": round(gbs, 1), "frac_mfma": round(f_m, 4),
                     "frac_hbm": round(f_h, 4), "frac": round(f_m if bound == "mfma" else f_h, 4)})
    return rows


def kernel_table(events, mfma_peak_tflops):
    """The GEMM-family operator launches of ONE step grouped by the kernel the library launched for them
    (diffsal_last_gemm_kernel(), recorded by ops at the call site): launches, HIP-event time, FLOPs actually issued, matrix-pipe
    rate.  Sorted by time: the first row is the dominant kernel."""
    agg = {}
    for e0, e1, fl, cls, nb, *rest in events:
        if cls not in GEMM_CLASSES:
            continue
        name = (rest[1] if len(rest) > 1 and rest[1] else f"({cls})").split(" [")[0]
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += fl
        a[2] += e0.elapsed_time(e1)
    rows = []
    for name in sorted(agg, key=lambda k: -agg[k][2]):
        n, fl, ms = agg[name]
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        rows.append({"name": name, "launches_per_step": n, "avg_us": round(ms * 1e3 / n, 2), "ms_per_step": round(ms, 4),
                     "gflop_per_step": round(fl / 1e9, 2), "tflops": round(tf, 2), "frac": round(tf / mfma_peak_tflops, 4)})
    return rows


def profiled_kernel(name, precision, bid):
    """rocprofv3 evidence for `name` from the newest profiles/rNN_{precision}_kernel_stats.csv / rNN_pmc_mfma_busy_{precision}.md of
    THIS build (profiles/rNN_manifest.jsonl lists the build of every artefact; other builds are refused)."""
    pdir = os.path.join(ROOT, "profiles")
    out = {"profile_avg_us": None, "mfma_busy": None, "profile_source": None}
    if not os.path.isdir(pdir) or bid is None:
        return out
    builds = {}
    for mf in sorted(f for f in os.listdir(pdir) if f.endswith("_manifest.jsonl")):
        for line in open(os.path.join(pdir, mf)):
            try:
                j = json.loads(line)
                builds[j.get("file")] = j.get("build")
            except Exception:  # noqa: BLE001
                pass
    key = name.split(" [")[0]
    # rocprofv3 leaves kernels with _Float16 / __bf16 template arguments mangled: match those on the base name + the type's mangling
    base = key.split("<")[0]
    tag = "DF16_" if "_Float16" in key else ("DF16b" if "__bf16" in key else None)

    def matches(text):
        return key in text or (tag is not None and f"{len(base)}{base}I" in text and tag in text)
    stats = sorted(f for f in os.listdir(pdir) if f.endswith(f"_{precision}_kernel_stats.csv"))
    busy = sorted(f for f in os.listdir(pdir) if f.endswith(f"_pmc_mfma_busy_{precision}.md"))
    notes = []
    if stats:
        f = stats[-1]
        if builds.get(f) == bid:
            import csv
            for row in csv.DictReader(open(os.path.join(pdir, f))):
                if matches(row.get("Name", "")):
                    out["profile_avg_us"] = round(float(row["AverageNs"]) / 1e3, 2)
                    break
            notes.append(f"profiles/{f}")
        else:
            notes.append(f"profiles/{f} is from build {builds.get(f)}, this is {bid}: refused")
    if busy:
        f = busy[-1]
        txt = open(os.path.join(pdir, f)).read()
        if f"build {bid}" in txt.splitlines()[0]:
            for line in txt.splitlines():
                if line.startswith("| `") and matches(line):
                    try:
                        out["mfma_busy"] = float(line.rstrip(" |").split("|")[-1])
                    except ValueError:
                        pass
                    break
            notes.append(f"profiles/{f}")
        else:
            notes.append(f"profiles/{f} is not from build {bid}: refused")
    out["profile_source"] = "; ".join(notes) if notes else None
    return out


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, the
    reference's own launch shape: `torchrun --nproc_per_node`, R/scripts/train_av.sh:13, R/train_av_data.py:38-61) and
    relay rank 0's JSON line.  The parent never initialises the GPU (no HIP call before or after the children start),
    so nothing is ever exec'ed or forked from a process that holds a device."""
    import socket
    import subprocess

    with socket.socket() as sk:      # a free port; the children bind it a moment later (a lost race fails the rendezvous loudly,
        sk.bind(("127.0.0.1", 0))     # and the watchdog below then ends every rank instead of hanging)
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread so that polling the children never blocks on a full pipe
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("DIFFSAL_BENCH_TIMEOUT_S", "3600"))
    failed = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            # a rank died (or the job overran): its peers would wait for it in init_process_group / a collective forever
            failed = abs(bad[0]) if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            sys.stderr.write(f"bench.py: rank failure or timeout (exit codes {codes}); all ranks stopped\n")
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return failed


def plumbing_train_leg(world):
    """CPU-only host: the gradient-exchange half of the `train` object on host tensors over gloo -- the same FlatParams /
    GradReducer code the GPU step uses (bucket hooks, asynchronous all_reduce, wait), a MIN / MAX checksum comparison of the
    replicas -- with every timing field null (there is no CPU training step to time)."""
    import numpy as np
    import torch.distributed as dist

    from diff_sal_amd.train_step import FlatParams, GradReducer

    torch.manual_seed(5)
    m = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.Tanh(), torch.nn.Linear(96, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8))
    flat = FlatParams(m, bucket_bytes=4 << 10)
    red = GradReducer(flat, None, exchange_single_rank=True, exchange=EXCHANGE_MODE)
    rank = dist.get_rank() if dist.is_initialized() else 0
    x = torch.randn(16, 64, generator=torch.Generator().manual_seed(100 + rank))
    # the reference draws ONE timestep per rank-batch from an unseeded generator (R/diffusion_trainer.py:111): every rank trains
    # at its own t0, the replicas must still end up identical (the gradient exchange is the only coupling)
    rng = np.random.RandomState(1000 + rank)
    for _ in range(3):
        t0 = int(rng.randint(0, 1000))
        flat.zero_grad()
        red.arm()
        (m(x) * (1.0 + t0 / 1000.0)).square().mean().backward()
        red.finish()
        flat.flat_p.add_(flat.flat_g, alpha=-0.1 / max(world, 1))
    chk = torch.stack([flat.flat_p.double().sum(), flat.flat_p.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    if dist.is_initialized():
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return {"what": "NONE: no GPU visible -- gradient-exchange plumbing only (toy module, gloo)", "value": None, "unit": "samples/s",
            "n_gpus": world, "ms_per_step": None, "grad_buckets": len(flat.buckets),
            "exchange": {"backend": dist.get_backend() if dist.is_initialized() else "none", "world": world,
                         "collective_executed": bool(dist.is_initialized() and red.exchange),
                         "bytes_per_step_per_rank": int(flat.numel * 4),
                         "buckets_mb": [round(len(r) * 4 / 2 ** 20, 4) for r in flat.buckets],
                         "mode": red.mode, "steps": 3, "rank_dependent_t0": True,
                         "launch_order": list(red.launch_order), "replicas_identical": bool(torch.equal(lo, hi)),
                         "allreduce_ms_per_bucket": None, "ms_per_step_without_exchange": None, "exposed_ms": None,
                         "overlap_frac": None}}


def plumbing_only(args, rank, world):
    """No GPU in this process (CPU-only host, e.g. the build container): exercise exactly the multi-rank plumbing of the
    benchmark -- rendezvous, barrier, MAX-over-ranks timing, rank-0 JSON -- over gloo, and say so.  The hot path itself has
    no CPU fallback, so `value` is null."""
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group(backend="gloo", init_method="env://")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    el = time.perf_counter() - t0
    ranks = 1
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        ranks = dist.get_world_size()
        dist.barrier()
    train = None
    if args.workload == "train" or not args.no_train_leg:
        train = plumbing_train_leg(world)
    if rank == 0 and args.workload == "train":
        print(json.dumps({"metric": "training samples/sec (diffusion train step of the denoiser: fwd + MSE + bwd + all-reduce + clip + Adam)",
                          "value": None, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": train["what"]}, "exchange": train["exchange"],
                          "rccl_ranks": ranks, "backend": "gloo", "devices": ["cpu"] * world, "max_over_ranks_s": round(el, 4)}),
              flush=True)
    elif rank == 0:
        print(json.dumps({"metric": "denoise-steps/sec (batch x NFE / wall time), 16x224x384 clip, 50-step DPM-Solver",
                          "train": train,
                          "value": None, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "NONE: no GPU visible -- launcher / rendezvous plumbing only (gloo)"},
                          "rccl_ranks": ranks, "backend": "gloo", "devices": ["cpu"] * world,
                          "max_over_ranks_s": round(el, 4)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


_REAL_STDOUT_FD = None


def quiet_stdout() -> None:
    """stdout carries ONE JSON line and nothing else: until `emit`, file descriptor 1 points at stderr, so that whatever a
    library writes to the C stdout (RCCL prints a version banner through printf when a communicator is created, and libc
    would flush it behind our line at exit) lands in the log instead."""
    global _REAL_STDOUT_FD
    if _REAL_STDOUT_FD is None:
        sys.stdout.flush()
        _REAL_STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result) -> None:
    """Flush everything that was written meanwhile (to stderr), give file descriptor 1 back and print the line."""
    global _REAL_STDOUT_FD
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    if _REAL_STDOUT_FD is not None:
        os.dup2(_REAL_STDOUT_FD, 1)
        os.close(_REAL_STDOUT_FD)
        _REAL_STDOUT_FD = None
    print(json.dumps(result), flush=True)
    quiet_stdout()          # teardown chatter (communicator destruction) goes to the log as well


class Config:
    """The decoder of R/cfgs/audio_visual.py:50-82 / R/cfgs/visual.py:33-70 (the only configuration the reference ships)."""
    img_size = (224, 384)
    up_channel = (768, 384, 192, 96)
    ori_embed_dim, down_embed_dim = 768, 96
    num_heads = (2, 2, 2, 2)
    kernel_kv = (2, 4, 8, 16)
    temporal_list = (5, 5, 5, 5)
    dilation = (0, 2, 2, 2)
    num_stages = 4


def synthetic_weights(net, seed=20240607):
    """Random-init weights of the reference architecture (no checkpoints offline): fan-in scaled normals for matrices and
    kernels, near-identity normalisation layers, non-trivial BatchNorm running statistics -- so activations keep O(1)
    magnitude through all four stages and no branch of the graph degenerates."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in net.state_dict().items():
        if not v.dtype.is_floating_point:
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1 and k.endswith("weight"):            # norm scales
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:                                      # biases
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5
    return sd


def build_net(cfg, device):
    from diff_sal_amd.sal_unet import SalUNet

    n = cfg.num_stages
    net = SalUNet(
        image_based=True, img_size=cfg.img_size, frames_len=1, mid_num_stages=n, temporal_size=9,
        temporal_list=list(cfg.temporal_list), futr_num_stages=0, ori_embed_dim=cfg.ori_embed_dim,
        down_embed_dim=cfg.down_embed_dim, patch_size=[0, 3, 3, 3], patch_stride=[0, 1, 1, 1],
        patch_padding=list(cfg.dilation), up_channel=list(cfg.up_channel), num_heads=list(cfg.num_heads),
        mlp_ratio=[2.0] * n, drop_path_rate=[0.15] * n, qkv_bias=[True] * n, kv_proj_method=["avg"] * n,
        kernel_kv=list(cfg.kernel_kv), padding_kv=[0] * n, stride_kv=list(cfg.kernel_kv),
        q_proj_method=["dw_bn"] * n, kernel_q=[3] * n, padding_q=[1] * n, stride_q=[1] * n)
    sd = synthetic_weights(net)
    net.load_state_dict(sd)
    return net.to(device).eval(), sd


class Top(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.decoder_net, self.audio_net, self.visual_net = net, None, None


EXCHANGE_MODE = "allreduce"     # --exchange: how the gradient buckets are summed over the ranks (train_step.GradReducer)


def build_train_step(cfg, net, feats, audio, dev, rank, *, batch, av, full, exchange_single_rank=False):
    """The model, inputs and DiffusionTrainStep of BASELINE configs[3] on this rank's synthetic clips."""
    import numpy as np

    from diff_sal_amd.train_step import DiffusionTrainStep

    B = batch
    H, W = cfg.img_size
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    sal = torch.rand((B, 1, H, W), generator=g).to(dev)
    if full:
        # the step as the reference runs it (R/diffusion_trainer.py:212-235): the encoders are INSIDE the step --
        # MViTv2-S forward + backward, VGGish forward (frozen, no_grad) and AudioAttnNet forward + backward in AV mode
        from diff_sal_amd.audio_attention import AudioAttnNet
        from diff_sal_amd.diff_model import VideoSaliencyModel
        from diff_sal_amd.mvit import MViT
        from diff_sal_amd.vggish import VGGish

        torch.manual_seed(11)                         # same random-init encoder weights on every rank
        enc = MViT(arch="small", out_scales=[0, 1, 2, 3])
        kw = dict(visual_net=enc, decoder_net=net)
        if av:
            kw.update(audio_net=VGGish(pretrained=False),
                      spatiotemp_net=AudioAttnNet(depth=1, heads=2, dim=512, mlp_dim=256, patch_dim=512, num_patches=16, height=7,
                                                  width=12, pool="cls", dim_head=64))
        model = VideoSaliencyModel(channel_list=None, **kw).to(dev)
        cond = {"img": torch.randn((B, 3, 16, H, W), generator=g).to(dev)}
        if av:
            cond["audio"] = torch.randn((B, 1, 9, H // 2, W // 2), generator=g).to(dev)
        ts = DiffusionTrainStep(model, exchange_single_rank=exchange_single_rank, exchange=EXCHANGE_MODE)
    else:
        cond = {"feat_list": feats, "audio_feat": audio}
        ts = DiffusionTrainStep(net, exchange_single_rank=exchange_single_rank, exchange=EXCHANGE_MODE)   # reference hyper-parameters: Adam 1e-4, clip 1.0, dropout 0.1
    ts._rng = np.random.RandomState(99)               # same timestep sequence on every rank / run
    return ts, sal, cond


def rank_table(own_seconds, units_per_rank, dev, world, unit):
    """Every rank's own rate over the timed region (its clock stops at its own device synchronize, before the closing barrier):
    the spread tells a straggling GPU from a uniformly slow job; `value` of the line stays units of all ranks / MAX time."""
    rates = [units_per_rank / own_seconds]
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([own_seconds], device=dev, dtype=torch.float64)
        out = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(out, tt)
        rates = [units_per_rank / float(o.item()) for o in out]
    return {"unit": unit + " per rank", "per_rank": [round(r, 3) for r in rates], "min": round(min(rates), 3), "max": round(max(rates), 3),
            "spread": round(max(rates) / min(rates) - 1.0, 4)}


def _dist_on():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def timed_train_region(ts, sal, cond, steps, dev, profile_last=False):
    """exactly `steps` training steps between barrier + synchronize on both sides; MAX over ranks.  Returns (seconds, last loss,
    events of the last step if profile_last)."""
    import torch.distributed as dist

    from diff_sal_amd import ops

    ev = []
    torch.cuda.synchronize()
    if _dist_on():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        if profile_last and i == steps - 1:
            ops.PROFILE = []
        loss = ts.step(sal, cond)
    if ops.PROFILE is not None:
        ev, ops.PROFILE = ops.PROFILE, None
    timed_train_region.host_issue_s = time.perf_counter() - t0      # the host has queued every launch of the region by now
    torch.cuda.synchronize()
    timed_train_region.own_s = time.perf_counter() - t0             # this rank's own clock (before the closing barrier)
    if _dist_on():
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if _dist_on():
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el, loss, ev


def exchange_report(ts, sal, cond, steps, dev, world, ms_with):
    """What the ONE collective of the training step costs and how much of it the backward hides (north_star: a single RCCL grad
    all-reduce over xGMI): payload, blocking time of each bucket's all-reduce measured alone, the step re-timed with the
    exchange switched off, and from the two the exposed time and the overlap fraction.  Also whether the replicas stayed
    bit-identical (MIN / MAX all-reduce of a parameter checksum) -- taken BEFORE the exchange is switched off."""
    import torch.distributed as dist

    flat, red = ts.flat, ts.reducer
    rep = {"backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if _dist_on() else "none",
           "world": world, "collective_executed": bool(_dist_on() and red.exchange),
           "mode": red.mode, "collectives_per_bucket": list(red.collectives),
           "bytes_per_step_per_rank": int(flat.numel * 4), "buckets_mb": [round(len(r) * 4 / 2 ** 20, 2) for r in flat.buckets]}
    chk = torch.stack([flat.flat_p.double().sum(), flat.flat_p.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    if _dist_on():
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    rep["replicas_identical"] = bool(torch.equal(lo, hi))
    rep["param_checksum"] = [float(chk[0].item()), float(chk[1].item())]
    if not (_dist_on() and red.exchange):
        rep.update(allreduce_ms_per_bucket=None, ms_per_step_without_exchange=None, exposed_ms=None, overlap_frac=None)
        return rep
    scratch = torch.zeros_like(flat.flat_g)
    per = []
    for r in flat.buckets:          # blocking all-reduce of each bucket, alone on the GPU: median of 5
        v = scratch[r.start:r.stop]
        ts_ = []
        for _ in range(6):
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=ts.group)
            torch.cuda.synchronize()
            ts_.append((time.perf_counter() - t0) * 1e3)
        per.append(round(median(ts_[1:]), 4))
    rep["allreduce_ms_per_bucket"] = per
    total = sum(per)
    gbytes = flat.numel * 4 / 1e9
    rep["allreduce_ms_total"] = round(total, 4)
    # ring all-reduce moves 2 (w-1)/w of the payload over each rank's links
    rep["allreduce_bus_gbs"] = round(gbytes * 2 * (world - 1) / max(world, 1) / (total * 1e-3), 2) if total > 0 and world > 1 else None
    red.exchange = False            # same step, no collective (the replicas drift apart from here on: last measurement)
    try:
        ts.broadcast_buffers, keep_bb = False, ts.broadcast_buffers
        ts.step(sal, cond)
        el, _, _ = timed_train_region(ts, sal, cond, steps, dev)
    finally:
        red.exchange, ts.broadcast_buffers = True, keep_bb
    ms_without = el / steps * 1e3
    exposed = max(0.0, ms_with - ms_without)
    rep["ms_per_step_without_exchange"] = round(ms_without, 4)
    rep["exposed_ms"] = round(exposed, 4)
    rep["overlap_frac"] = round(max(0.0, min(1.0, 1.0 - exposed / total)), 4) if total > 0 else None
    # weak-scaling efficiency of the step against the same step without its one collective (the N = 1 figure on this node)
    rep["efficiency_vs_no_exchange"] = round(ms_without / ms_with, 4) if ms_with > 0 else None
    rep["modes"] = exchange_modes(ts, sal, cond, steps, dev, world, ms_without)
    return rep


def exchange_modes(ts, sal, cond, steps, dev, world, ms_without):
    """The same step under BOTH exchange forms in one invocation -- ring all-reduce per bucket, and reduce-scatter + all-gather
    (all seven xGMI links of a GPU instead of one per ring hop) -- so that one scaling run A/Bs them: step time, the collectives'
    blocking time measured alone, bus bandwidth (2 (w - 1) / w of the payload per rank for either form), exposed time and overlap."""
    import torch.distributed as dist

    flat, red = ts.flat, ts.reducer
    out = {}
    keep = red.mode
    gbytes = flat.numel * 4 / 1e9
    try:
        for mode in ("allreduce", "reduce_scatter"):
            red.mode = mode
            ts.step(sal, cond)
            el, _, _ = timed_train_region(ts, sal, cond, steps, dev)
            ms = el / steps * 1e3
            scratch = torch.zeros_like(flat.flat_g)
            tot = 0.0
            for r in flat.buckets:
                v = scratch[r.start:r.stop]
                n = v.numel()
                ts_ = []
                for _ in range(4):
                    torch.cuda.synchronize()
                    dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if mode == "reduce_scatter" and n % world == 0 and red._native_rs:
                        mine = v[red.rank * (n // world):(red.rank + 1) * (n // world)]
                        dist.reduce_scatter_tensor(mine, v, op=dist.ReduceOp.SUM, group=ts.group)
                        dist.all_gather_into_tensor(v, mine, group=ts.group)
                    else:
                        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=ts.group)
                    torch.cuda.synchronize()
                    ts_.append((time.perf_counter() - t0) * 1e3)
                tot += median(ts_[1:])
            exposed = max(0.0, ms - ms_without)
            out[mode] = {"ms_per_step": round(ms, 4), "collectives_ms_total": round(tot, 4),
                         "bus_gbs": round(gbytes * 2 * (world - 1) / max(world, 1) / (tot * 1e-3), 2) if tot > 0 and world > 1 else None,
                         "exposed_ms": round(exposed, 4),
                         "overlap_frac": round(max(0.0, min(1.0, 1.0 - exposed / tot)), 4) if tot > 0 else None,
                         "collectives_per_bucket": sorted(set(red.collectives))}
    finally:
        red.mode = keep
    return out


def train_leg(cfg, dev, rank, world, *, steps=10, warmup=3, batch=4):
    """The `train` object of the default bench line (every N): BASELINE configs[3] -- full-model audio-visual training step,
    per-GPU batch 4, weak scaling -- so that one `bench.py --gpus N` run also answers north_star's multi-GPU training
    question.  At N = 1 a single-rank RCCL group is created so that the collective code path (bucket hooks -> asynchronous
    all_reduce -> buffer broadcast) executes on hardware too."""
    import torch.distributed as dist

    own_group = False
    note = None
    if not _dist_on():
        try:
            import socket
            from datetime import timedelta

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                    timeout=timedelta(minutes=5))
            own_group = True
        except Exception as e:  # noqa: BLE001
            note = f"single-rank RCCL group could not be created ({type(e).__name__}: {e}); step timed without a collective"
    try:
        net, _ = build_net(cfg, dev)
        H, W = cfg.img_size
        ts, sal, cond = build_train_step(cfg, net, None, None, dev, rank, batch=batch, av=True, full=True,
                                         exchange_single_rank=True)
        for _ in range(max(warmup, 1)):
            ts.step(sal, cond)
        el, loss, _ = timed_train_region(ts, sal, cond, steps, dev)
        ranks_tab = rank_table(timed_train_region.own_s, batch * steps, dev, world, "samples/s")
        ms = el / steps * 1e3
        out = {"what": "BASELINE configs[3]: audio-visual training step of the WHOLE model (MViTv2-S fwd+bwd, frozen VGGish, "
                       "AudioAttnNet fwd+bwd, SalUNet fwd+bwd; MSE, one bucketed gradient all-reduce, clip 1.0, Adam), "
                       f"per-GPU batch {batch} (global {world * batch}), 224x384, fp32, synthetic data; not the headline metric",
               "value": round(world * batch * steps / el, 3), "unit": "samples/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": round(ms, 4), "scaling": "weak", "batch_per_gpu": batch,
               "trainable_params": ts.flat.live_numel, "grad_buckets": len(ts.flat.buckets), "final_loss": float(loss.item())}
        out["ranks"] = ranks_tab
        out["exchange"] = exchange_report(ts, sal, cond, steps, dev, world, ms)
        if note:
            out["note"] = note
        return out
    finally:
        if own_group and _dist_on():
            dist.destroy_process_group()


def bench_train(args, net, cfg, feats, audio, dev, rank, world):
    """BASELINE configs[3]: the diffusion training step of the denoiser on a per-GPU batch (weak scaling)."""
    B, av = args.batch, args.mode == "av"
    full = args.train_scope == "full"
    ts, sal, cond = build_train_step(cfg, net, feats, audio, dev, rank, batch=B, av=av, full=full)
    for _ in range(max(args.warmup, 1)):
        ts.step(sal, cond)
    regions, ev, host_issue = [], [], []
    for rep_i in range(args.repeats):      # each region: exactly K steps between barrier + synchronize; MAX over ranks
        el, loss, e = timed_train_region(ts, sal, cond, args.steps, dev, profile_last=rep_i == args.repeats - 1)
        if rep_i == 0:
            ranks_tab = rank_table(timed_train_region.own_s, B * args.steps, dev, world, "samples/s")
        ev = e or ev
        regions.append(el)
        host_issue.append(timed_train_region.host_issue_s)
    elapsed = median(regions)
    if args.dump_launches and rank == 0:     # per-launch table of the profiled step (shape -> TF/s, GB/s)
        rows = []
        for e0, e1, fl, cls, nb, *note in ev:
            us = e0.elapsed_time(e1) * 1e3
            rows.append({"class": cls, "op": note[0] if note else "", "kernel": note[1] if len(note) > 1 else "", "gflop": round(fl / 1e9, 4), "mbytes": round(nb / 1e6, 3),
                         "us": round(us, 2), "tflops": round(fl / us / 1e6, 2) if us > 0 else 0.0,
                         "gbs": round(nb / us / 1e3, 1) if us > 0 else 0.0})
        json.dump({"workload": "train", "mode": args.mode, "batch": B, "launches": rows}, open(args.dump_launches, "w"), indent=0)
    gemm_ev = [e for e in ev if e[3] in GEMM_CLASSES]
    k_ms = sum(e[0].elapsed_time(e[1]) for e in gemm_ev)
    k_flops = sum(e[2] for e in gemm_ev)
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    ranks_seen, devices = 1, [torch.cuda.get_device_name(dev)]
    if world > 1:
        import torch.distributed as dist

        ranks_seen = dist.get_world_size()
        devices = [None] * world
        dist.all_gather_object(devices, f"cuda:{dev.index} {torch.cuda.get_device_name(dev)}")
    result = {
        "metric": "training samples/sec (diffusion train step of the denoiser: fwd + MSE + bwd + all-reduce + clip + Adam)",
        "value": round(world * B * args.steps / elapsed, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: " + ("audio-visual" if av else "visual-only")
                   + f" training step, per-GPU batch {B} (global {world * B}), 224x384, "
                   + ("VideoSaliencyModel end to end: MViTv2-S (fwd+bwd)" + (" + VGGish (frozen) + AudioAttnNet (fwd+bwd)" if av else "")
                      + " + SalUNet (fwd+bwd), as R/diffusion_trainer.py:212-235 runs it" if full else
                      "SalUNet parameters only (conditioning features are inputs; --train-scope decoder)"),
                   "train_scope": args.train_scope,
                   "batch_per_gpu": B, "trainable_params": ts.flat.live_numel, "grad_buckets": len(ts.flat.buckets),
                   "exchange": "RCCL all-reduce of the flat fp32 gradient, bucketed, overlapped with backward",
                   "final_loss": float(loss.item())},
        "ranks": ranks_tab,
        "exchange": exchange_report(ts, sal, cond, args.steps, dev, world, elapsed / args.steps * 1e3),
        "repeats": args.repeats, "ms_per_step_all_regions": [round(r / args.steps * 1e3, 4) for r in regions],
        # host time to QUEUE a step's launches (Python + autograd tape + ~1 800 launches): when it reaches ms_per_step the step is
        # issue-bound and kernel-side savings no longer show (the first, unprofiled regions are the ones to read)
        "host_issue_ms_per_step": [round(h / args.steps * 1e3, 4) for h in host_issue],
        "rccl_ranks": ranks_seen, "backend": "nccl (RCCL)" if world > 1 else "none (single rank)", "devices": devices,
        "roofline": {"kernel": "diffsal::igemm_kernel + wgrad_kernel (fp32 MFMA: forward, data-gradient and "
                               "weight-gradient convolutions/GEMMs of one step)",
                     "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None, "launches_per_step": len(gemm_ev),
                     "step_ms_in_kernel": round(k_ms, 3), "classes": class_table(ev, FP32_MFMA_PEAK_TFLOPS),
                     # how much of the step the class table brackets (HIP events of every library operator of one step; the rest
                     # is torch-native elementwise / copy / fill kernels of the autograd tape and launch gaps)
                     "classified_ms": round(sum(e[0].elapsed_time(e[1]) for e in ev), 3), "operators_bracketed": len(ev)},
    }
    if rank == 0:
        emit(result)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--mode", choices=["vis", "av"], default="vis", help="vis = BASELINE configs[1]; av = configs[2]")
    ap.add_argument("--workload", choices=["sample", "train"], default="sample",
                    help="sample = the headline metric; train = BASELINE configs[3] (one step = prepare_data + forward + "
                         "MSE + backward + gradient all-reduce + clip + Adam on a per-GPU batch), reported as samples/s")
    ap.add_argument("--no-reference-graph", action="store_true",
                    help="skip the extra profiled step on the reference's own graph (profiling passes: keeps the trace to the shipped step)")
    ap.add_argument("--tap16", default=None, help="16-bit storage: comma list of s1,s2,s3,mt uses of the tap form (tuning aid)")
    ap.add_argument("--train-scope", choices=["full", "decoder"], default="full",
                    help="--workload train: full = MViT + (VGGish, AudioAttnNet) + SalUNet inside the step, as the reference; "
                         "decoder = the denoiser alone on given features (round-1 measurement)")
    ap.add_argument("--sampler-mode", choices=["auto", "eager", "graph", "f1"], default="auto",
                    help="auto = what DiffusionSampler does by default: eager from 3 clips per step on (the headline batch of 4), "
                         "HIP-graph replay of whole trajectories at 1-2 clips; eager / graph force one; f1 = step-invariant "
                         "shortcut of visual-only mode (1 evaluation per trajectory), reported separately")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16", "fp16"], default="fp32",
                    help="fp32 = headline (exact fp32 MFMA); bf16x3 = split-precision bf16 MFMA on fp32 tensors (~4e-6); "
                         "bf16 / fp16 = 16-bit STORAGE of activations + packed weights, native 16-bit MFMA, fp32 accumulate "
                         "(BASELINE configs[1] / configs[4]; own tolerance table, DESIGN.md 2b).  All but fp32 are opt-in "
                         "modes reported separately from the headline")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the extra reduced-precision passes reported beside the headline")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of exactly --steps steps each; the reported value is their median (SURVEY 8d)")
    ap.add_argument("--dump-launches", default=None,
                    help="write every operator launch of the profiled step (class, GFLOP, MB, us, TF/s, GB/s) to this JSON file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=["allreduce", "reduce_scatter"], default="allreduce",
                    help="training legs: gradient buckets summed by one ring all-reduce each (default) or by reduce-scatter + "
                         "all-gather (same result; for A/B on an xGMI node)")
    ap.add_argument("--no-cpu-trajectory", action="store_true",
                    help="cpu_baseline without its one 50-NFE single-clip trajectory (~10-15 s of CPU time)")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the `train` object (BASELINE configs[3]: ~10 full-model AV training steps per rank, with the "
                         "gradient exchange report) that the default line carries at every N")
    ap.add_argument("--train-leg-steps", type=int, default=10)
    ap.add_argument("--no-encoders", action="store_true", help="skip the end-to-end (encoders + 50 steps) leg")
    ap.add_argument("--no-solo-leg", action="store_true",
                    help="N > 1: skip rank 0's solo timing of the same steps (the N = 1 reference of `scaling_efficiency`)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B aid: set a SalUNet switch (fuse_resblock=0, up_commute=0, ...) on the benchmarked network; recorded in config")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads for the CPU baseline (32 is the fastest setting measured on the 256-core box)")
    args = ap.parse_args()
    global EXCHANGE_MODE
    EXCHANGE_MODE = args.exchange

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("DIFFSAL_BENCH_FAULT_RANK") == str(rank) and world > 1:
        # test hook (tests/test_host_surface.py): this rank dies before the rendezvous; spawn_ranks must end its peers
        sys.exit(3)
    if not torch.cuda.is_available():
        return plumbing_only(args, rank, world)
    quiet_stdout()        # a rank process: from here on only `emit` writes to the real stdout
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", init_method="env://")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from diff_sal_amd import ops
    from diff_sal_amd.sampling import DiffusionSampler

    cfg = Config()
    B, av = args.batch, args.mode == "av"
    net, sd = build_net(cfg, dev)
    STORAGE = {"bf16": torch.bfloat16, "fp16": torch.float16}

    def set_precision(mode):
        net.compute_dtype = STORAGE.get(mode, torch.float32)
        net.gemm_precision = "bf16x3" if mode == "bf16x3" else "fp32"

    set_precision(args.precision)
    for kv in args.set:
        name, _, val = kv.partition("=")
        if not hasattr(net, name):
            raise SystemExit(f"--set {kv}: SalUNet has no switch {name!r}")
        cur = getattr(net, name)
        setattr(net, name, (val not in ("0", "false", "False", "")) if isinstance(cur, bool) else type(cur)(val))
    if args.tap16 is not None:
        net.tap_conv16 = tuple(x for x in args.tap16.split(",") if x)

    # synthetic clips, resident in HBM before the timed region; rank-dependent seed (each rank owns its clips)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    H, W = cfg.img_size
    x_T = torch.randn((B, 1, H, W), generator=g).to(dev)
    feats = [torch.randn((B, c, 8, H // s, W // s), generator=g).to(dev)
             for c, s in zip(cfg.up_channel, (32, 16, 8, 4))]
    audio = torch.randn((B, 512, 9, H // 32, W // 32), generator=g).to(dev) if av else None

    if args.workload == "train":
        return bench_train(args, net, cfg, feats, audio, dev, rank, world)

    sampler = DiffusionSampler(Top(net), timesteps=NFE_PER_TRAJECTORY, sample_type="dpmsolver", skip_type="logSNR",
                               denoise=True, training_target="x0")

    state = {"budget": 0, "profile_last": False}
    inner = net.forward

    def counted(x, t, f, a=None):
        if state["budget"] <= 0:
            raise _Stop
        state["budget"] -= 1
        if state["budget"] == 0 and state["profile_last"]:
            ops.PROFILE = []
            try:
                return inner(x, t, f, a)
            finally:
                state["events"], ops.PROFILE = ops.PROFILE, None
        return inner(x, t, f, a)

    net.forward = counted
    inner_fused = net.forward_fused_update

    def counted_fused(x, t, f, a=None, **kw):
        """the sampler's fused step (network evaluation + solver update in one call) counts as one step, like `counted`"""
        if state["budget"] <= 0:
            raise _Stop
        state["budget"] -= 1
        if state["budget"] == 0 and state["profile_last"]:
            ops.PROFILE = []
            try:
                return inner_fused(x, t, f, a, **kw)
            finally:
                state["events"], ops.PROFILE = ops.PROFILE, None
        return inner_fused(x, t, f, a, **kw)

    net.forward_fused_update = counted_fused

    if args.sampler_mode == "auto":
        args.sampler_mode = "graph" if B <= 2 else "eager"
        if args.sampler_mode == "graph" and args.steps % NFE_PER_TRAJECTORY != 0:      # whole trajectories only
            args.steps = max(NFE_PER_TRAJECTORY, args.steps // NFE_PER_TRAJECTORY * NFE_PER_TRAJECTORY)
    sampler.hip_graph = False
    special = args.sampler_mode != "eager"
    if special:
        assert args.steps % NFE_PER_TRAJECTORY == 0, "graph / f1 modes time whole 50-NFE trajectories"
        net.forward, net.forward_fused_update = inner, inner_fused
        sampler.hip_graph = args.sampler_mode == "graph"
        sampler.step_invariant_shortcut = args.sampler_mode == "f1"

    def run_steps(n, profile_last=False):
        if special:
            for _ in range(max(1, n // NFE_PER_TRAJECTORY)):
                sampler.sample_dpm_solver(x_T, feats, audio)
            return
        state["budget"], state["profile_last"] = n, profile_last
        while state["budget"] > 0:
            try:
                sampler.sample_dpm_solver(x_T, feats, audio)
            except _Stop:
                pass

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    def timed_region(profile_last):
        """exactly K steps between barrier + synchronize on both sides; MAX over ranks"""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, profile_last=profile_last)
        torch.cuda.synchronize()
        state["own_s"] = time.perf_counter() - t0       # this rank's own clock (before the closing barrier)
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist

            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    run_steps(max(args.warmup, 1))
    # N > 1: rank 0 first times the same K steps ALONE (its peers wait at the barrier, their GPUs idle): the N = 1 value of this very
    # run configuration, against which the line reports its own weak-scaling efficiency (the driver computes its own from the
    # per-N lines; this one needs no second invocation)
    solo = None
    if world > 1 and not args.no_solo_leg:
        torch.cuda.synchronize()
        barrier()
        if rank == 0:
            t0 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            solo = time.perf_counter() - t0
        barrier()
    regions = []
    for i in range(args.repeats):
        regions.append(timed_region(i == args.repeats - 1))
        if i == 0:
            ranks_tab = rank_table(state["own_s"], B * args.steps, dev, world, "denoise-steps/s")
    elapsed = median(regions)

    # ---- roofline of the dominant kernel, from the HIP events recorded inside the timed region ----
    all_ev = state.get("events") or []
    ev = [e for e in all_ev if e[3] in GEMM_CLASSES]          # the implicit-GEMM family = the dominant kernel
    k_ms = sum(e[0].elapsed_time(e[1]) for e in ev)
    k_flops = sum(e[2] for e in ev)
    n_launch = max(len(ev), 1)
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    if args.dump_launches and rank == 0:
        rows = []
        for e0, e1, fl, cls, nb, *note in all_ev:
            us = e0.elapsed_time(e1) * 1e3
            rows.append({"class": cls, "op": note[0] if note else "", "kernel": note[1] if len(note) > 1 else "", "gflop": round(fl / 1e9, 4), "mbytes": round(nb / 1e6, 3), "us": round(us, 2),
                         "tflops": round(fl / us / 1e6, 2) if us > 0 else 0.0, "gbs": round(nb / us / 1e3, 1) if us > 0 else 0.0})
        json.dump({"precision": args.precision, "mode": args.mode, "batch": B, "launches": rows}, open(args.dump_launches, "w"), indent=0)
    peak_for_mode = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
    # roofline.traffic: HBM bytes of the GEMM family per launch from the PMC passes of THIS code (tools/profile_round.sh; separate
    # --pmc runs as the guide prescribes).  A profile taken on other sources is refused: traffic = null, and the line says why.
    traffic, traffic_src = None, None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from build_id import source_id
        bid = source_id()
    except Exception:  # noqa: BLE001
        bid = None
    # artefact suffix as tools/profile_round.sh writes it: precision [_b<batch> when not the headline batch of 4] [_av]
    suf = args.precision + (f"_b{B}" if B != 4 else "") + ("_av" if av else "")
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(f"_hbm_traffic_{suf}.json")) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if cands:
        tf = os.path.join(ROOT, "profiles", cands[-1])
        try:
            tj = json.load(open(tf))
            if bid is not None and tj.get("build") == bid:
                # HBM bytes of the whole GEMM family per step / its operator launches per step: per launch, like `achieved`
                traffic = tj["bytes_per_step"] / n_launch
                traffic_src = f"profiles/{cands[-1]} (build {bid})"
            else:
                traffic_src = f"profiles/{cands[-1]} is from build {tj.get('build')}, this is {bid}: refused"
        except Exception as e:  # noqa: BLE001
            traffic_src = f"profiles/{cands[-1]} unreadable ({e})"
    common = {"launches_per_step": n_launch, "avg_launch_us": round(k_ms * 1e3 / n_launch, 2),
              "flops_per_launch": k_flops / n_launch, "step_ms_in_kernel": round(k_ms, 3), "traffic_source": traffic_src,
              "step_ms_all_kernels": round(sum(e[0].elapsed_time(e[1]) for e in all_ev), 3),
              "classes": class_table(all_ev, peak_for_mode)}
    ref_graph = None
    if args.precision == "fp32" and getattr(net, "tap_conv", False) and not special and not args.no_reference_graph:
        # the same step on the reference's own graph (3x3 convolutions AFTER the bilinear up-samplings, sal_unet.py:480-489,
        # common_block.py:196-216): its GEMM FLOPs are SURVEY 8(d)'s algorithmic figure; the shipped path executes fewer
        net.tap_conv = False
        had_wino, net.winograd = getattr(net, "winograd", False), False     # ... and every 3x3 convolution on the direct kernel
        try:
            run_steps(3, profile_last=True)     # outside every timed region; the third step is the profiled one
            torch.cuda.synchronize()
            ref_ev = [e for e in (state.get("events") or []) if e[3] in GEMM_CLASSES]
            ref_flops = sum(e[2] for e in ref_ev)
            ref_ms = sum(e[0].elapsed_time(e[1]) for e in ref_ev)
        finally:
            net.tap_conv = True
            net.winograd = had_wino
        # the streaming kernels that belong to those layers in the shipped path: tap gathers, Winograd F(4x4) transforms
        tap_ms = sum(e[0].elapsed_time(e[1]) for e in all_ev if e[3] in ("K12-tap", "K14-tap") or e[3].endswith("-xf"))
        eq = ref_flops / ((k_ms + tap_ms) * 1e-3) / 1e12
        ref_graph = {"gemm_gflop_per_step": round(ref_flops / 1e9, 1), "executed_gemm_gflop_per_step": round(k_flops / 1e9, 1),
                     "ms_gemm_plus_tap_gathers": round(k_ms + tap_ms, 3), "ms_tap_gathers_and_winograd_transforms": round(tap_ms, 3), "tflops_equivalent": round(eq, 2),
                     "frac_equivalent": round(eq / FP32_MFMA_PEAK_TFLOPS, 4),
                     "direct_graph_ms_in_kernel": round(ref_ms, 3),
                     "direct_graph_tflops": round(ref_flops / (ref_ms * 1e-3) / 1e12, 2) if ref_ms > 0 else None,
                     "note": "direct_graph_* = the reference's own graph on the same kernels (tap_conv off), one profiled step outside "
                             "the timed regions; *_equivalent prices its GEMM FLOPs at the time the shipped path needs for them"}
    ktab = kernel_table(all_ev, peak_for_mode)
    dominant = None
    if ktab:
        dominant = dict(ktab[0])
        dominant["share_of_step_in_kernel"] = round(dominant["ms_per_step"] / max(common["step_ms_all_kernels"], 1e-9), 4)
        dominant.update(profiled_kernel(dominant["name"], suf, bid))
        dominant["note"] = ("launches / avg_us / tflops measured live with HIP events around the operator launches of one timed step (the library "
                            "names the kernel it launched: diffsal_last_gemm_kernel); profile_avg_us / mfma_busy are read from the same-build "
                            "rocprofv3 artefacts under profiles/ when they exist (SQ_VALU_MFMA_BUSY_CYCLES / all SIMD cycles)")
    common["kernels"] = ktab
    if args.precision == "fp32":
        # roofline.achieved / frac = the matrix-pipe rate on the FLOPs the GEMM family actually ISSUES in one step (every
        # implicit-GEMM, Winograd, DMA-GEMM and fused-block launch; HIP-event time of exactly those launches): a kernel figure,
        # never above 1.  The reference graph's GEMM FLOPs priced at the shipped time (the contract's "algorithmic" figure; the
        # shipped step issues well under half of them: tap GEMMs at the source resolution, Winograd F(4x4,3x3), composed conv_in) stand
        # beside it as achieved_reference_graph / frac_reference_graph and may pass 1.
        roofline = {
            "kernel": "fp32 MFMA GEMM family of one step: diffsal::gemm_dma_kernel / igemm_kernel / igemm_linear_kernel / wino_gemm_kernel / "
                      "lin_stream_kernel / mlp_block_kernel / block_front_kernel (3x3 convs, token GEMMs, tap GEMMs, ReduceTemp, fused "
                      "transformer-block halves); per kernel: `kernels`, the largest: `dominant_kernel`",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "executed_mfma_tflops": round(achieved, 2), "executed_frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "dominant_kernel": dominant,
            "note": "achieved = FLOPs actually issued by the GEMM-family launches of one step / HIP-event time of exactly those launches "
                    "(kernel quality; <= 1 by construction).  *_reference_graph = the reference graph's GEMM FLOPs (SURVEY 8d) priced at the "
                    "time the shipped launches that implement those layers need (GEMM family + tap gathers): an end-to-end figure that can "
                    "pass 1 because about half of the reference graph's products are never issued",
            **common}
        if ref_graph is not None:
            roofline["achieved_reference_graph"] = ref_graph["tflops_equivalent"]
            roofline["frac_reference_graph"] = ref_graph["frac_equivalent"]
            roofline["reference_graph"] = ref_graph
    elif args.precision == "bf16x3":   # three bf16 MFMAs per fp32-accurate product: the instruction-level peak is the bf16 one
        roofline = {
            "kernel": "diffsal::igemm_kernel<..., bf16x3> (split-precision bf16 MFMA implicit GEMM, fp32 accumulate)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "executed_mfma_tflops": round(3 * achieved, 2), "executed_frac": round(3 * achieved / BF16_MFMA_PEAK_TFLOPS, 4),
            "note": "achieved = algorithmic 2MNK FLOPs / time; the kernel issues 3 bf16 MFMAs per product (hi*hi+hi*lo+lo*hi)",
            **common}
    else:
        roofline = {
            "kernel": f"diffsal::conv16_dma_kernel / gemm16_dma2_kernel / gemm_dma_kernel / igemm16_kernel / block16_kernel / block_front_kernel <{args.precision}> "
                      f"(native 16-bit MFMA GEMM family on {args.precision} storage, fp32 accumulate)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "dominant_kernel": dominant, **common}

    DT_LABEL = {"fp32": "f32", "bf16x3": "f32 in/out, bf16x3 split-precision MFMA (NOT the headline configuration)",
                "bf16": "bf16 storage, f32 accumulate (NOT the headline configuration; BASELINE configs[1] as written)",
                "fp16": "f16 storage, f32 accumulate (NOT the headline configuration; BASELINE configs[4] arithmetic)"}
    devices = [torch.cuda.get_device_name(dev)]
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist

        ranks_seen = dist.get_world_size()
        names = [None] * world
        dist.all_gather_object(names, f"cuda:{local_rank} {torch.cuda.get_device_name(dev)}")
        devices = names
    result = {
        "metric": "denoise-steps/sec (batch x NFE / wall time), 16x224x384 clip, 50-step DPM-Solver",
        "value": round(world * B * args.steps / elapsed, 3),
        "unit": "denoise-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "repeats": args.repeats, "ms_per_step_all_regions": [round(r / args.steps * 1e3, 4) for r in regions],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DT_LABEL[args.precision],
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]: DHF1k visual-only" if not av else "BASELINE configs[2]: AVAD audio-visual")
            + f", batch={B}/GPU, 224x384, 50-NFE DPM-Solver (multistep-2, logSNR, denoise-to-zero), faithful full graph",
            "batch_per_gpu": B, "nfe_per_trajectory": NFE_PER_TRAJECTORY, "sharding": "clips by rank, no collective",
            "step": "one SalUNet evaluation + DPM-Solver update on one batch",
            "sampler_mode": args.sampler_mode + ("" if not special else " (NOT the headline configuration)"),
            "gflop_per_clip_step": 151.61 if not av else 152.73,
            **({"switches (A/B run, NOT the shipped defaults)": list(args.set)} if args.set else {}),
        },
        "rccl_ranks": ranks_seen, "backend": "nccl (RCCL)" if world > 1 else "none (single rank)", "devices": devices,
        "ranks": ranks_tab,
        **({"solo_rank0": {"what": "rank 0 alone, same configuration and K steps, the other GPUs idle (N = 1 reference of this run)",
                           "value": round(B * args.steps / solo, 3), "ms_per_step": round(solo / args.steps * 1e3, 4)},
            "scaling_efficiency": round((world * B * args.steps / elapsed) / (world * B * args.steps / solo), 4)}
           if solo is not None else {}),
        "build": bid,
        "roofline": roofline,
    }

    if world == 1 and args.precision == "fp32" and not special and not args.no_alt_precision:
        # The same K steps once more in each opt-in reduced-precision mode, reported next to the headline (not instead of
        # it), with the output difference from the fp32 arithmetic on one network evaluation.
        with torch.no_grad():
            t_probe = torch.full((B,), 500, device=dev)
            y32 = inner(x_T, t_probe, feats, audio)
        alts = []
        for mode, desc in (
                ("bf16x3", "fp32 tensors; operands split into bf16 hi+lo inside the GEMM kernel, 3 bf16 MFMAs per product, fp32 "
                           "accumulation (diffsal_conv_desc.precision = 1)"),
                ("bf16", "bf16 STORAGE of activations and packed weights, native bf16 MFMA, fp32 accumulation and statistics "
                         "(SalUNet(compute_dtype=torch.bfloat16); BASELINE configs[1] as written)"),
                ("fp16", "fp16 STORAGE of activations and packed weights, native fp16 MFMA, fp32 accumulation and statistics "
                         "(SalUNet(compute_dtype=torch.float16); the arithmetic of BASELINE configs[4])")):
            set_precision(mode)
            with torch.no_grad():
                y3 = inner(x_T, t_probe, feats, audio)
            run_steps(max(args.warmup, 1))
            torch.cuda.synchronize()
            a0 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            a_el = time.perf_counter() - a0
            # 16-bit modes finish a step faster than Python can enqueue it: also time whole trajectories replayed from
            # a HIP graph (same kernels, same order; the sampler's optional mode)
            graph = None
            if mode in STORAGE and args.steps >= NFE_PER_TRAJECTORY:
                net.forward, net.forward_fused_update = inner, inner_fused
                gs = DiffusionSampler(Top(net), timesteps=NFE_PER_TRAJECTORY, sample_type="dpmsolver", skip_type="logSNR",
                                      denoise=True, training_target="x0", hip_graph=True)
                gs.sample_dpm_solver(x_T, feats, audio)
                torch.cuda.synchronize()
                ntraj = max(1, args.steps // NFE_PER_TRAJECTORY)
                g0 = time.perf_counter()
                for _ in range(ntraj):
                    gs.sample_dpm_solver(x_T, feats, audio)
                torch.cuda.synchronize()
                g_el = time.perf_counter() - g0
                graph = {"value": round(B * ntraj * NFE_PER_TRAJECTORY / g_el, 3), "unit": "denoise-steps/s",
                         "ms_per_step": round(g_el / (ntraj * NFE_PER_TRAJECTORY) * 1e3, 4),
                         "note": "whole 50-NFE trajectories replayed from one HIP graph"}
                net.forward, net.forward_fused_update = counted, counted_fused
            alts.append({"mode": mode, "what": desc, "value": round(B * args.steps / a_el, 3), "unit": "denoise-steps/s",
                         "ms_per_step": round(a_el / args.steps * 1e3, 4), "speedup_vs_headline": round(elapsed / a_el, 3),
                         "hip_graph": graph,
                         "max_abs_output_diff_vs_fp32": float((y3 - y32).abs().max().item()),
                         "output_range": [float(y32.min().item()), float(y32.max().item())]})
        set_precision("fp32")
        result["alt_precision"] = alts
    if world == 1 and args.precision in ("fp32", "bf16", "fp16") and not special and not args.no_encoders:
        # "also report end-to-end clips/s including encoders separately".  The once-per-clip encoders (MViTv2-S
        # video encoder; VGGish + AudioAttnNet in AV mode) on the HIP path, random-init weights of the reference
        # architectures, timed on a synthetic clip batch; end-to-end = encoders once + 50 denoising steps per clip batch.
        from diff_sal_amd.audio_attention import AudioAttnNet
        from diff_sal_amd.mvit import MViT
        from diff_sal_amd.vggish import VGGish

        torch.manual_seed(7)
        # 16-bit storage runs: the video encoder keeps its token stream and GEMM weights in the same storage type (MViT(compute_dtype))
        enc = MViT(arch="small", out_scales=[0, 1, 2, 3], compute_dtype=STORAGE.get(args.precision, torch.float32))
        enc = enc.to(dev).eval().requires_grad_(False)
        clip = torch.randn((B, 3, 16, H, W), device=dev)
        mods = [("mvit_s", lambda: enc(clip))]
        if av:
            vgg = VGGish(pretrained=False).to(dev).eval().requires_grad_(False)
            aan = AudioAttnNet(depth=1, heads=2, dim=512, mlp_dim=256, patch_dim=512, num_patches=16, height=7, width=12,
                               pool="cls", dim_head=64).to(dev).eval().requires_grad_(False)
            wav = torch.randn((B, 1, 9, 112, 192), device=dev)
            top = Top(net)
            from diff_sal_amd.diff_model import VideoSaliencyModel

            vm = VideoSaliencyModel(channel_list=None, audio_net=vgg, spatiotemp_net=aan)
            mods.append(("vggish+audio_attn", lambda: vm.forward_vggish(wav)))
        enc_ms = {}
        with torch.no_grad():
            for nm, fn in mods:
                fn()
                torch.cuda.synchronize()
                ts_ = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    ts_.append(time.perf_counter() - t0)
                enc_ms[nm] = round(median(ts_) * 1e3, 3)
        denoise_ms = elapsed / args.steps * 1e3 * NFE_PER_TRAJECTORY
        total_ms = denoise_ms + sum(enc_ms.values())
        result["end_to_end"] = {
            "what": "encoders once per clip batch + 50 denoising steps, all on the HIP path (not the headline metric)",
            "encoder_ms_per_batch": enc_ms, "denoise_ms_per_batch": round(denoise_ms, 3),
            "clips_per_s": round(B / (total_ms * 1e-3), 3), "encoder_share": round(sum(enc_ms.values()) / total_ms, 4)}
    if args.precision == "fp32" and not special and not args.no_train_leg:
        # every rank takes part (the gradient all-reduce is the one collective of the repo); the sampling network and its
        # inputs are released first
        net.forward, net.forward_fused_update = inner, inner_fused
        try:
            result["train"] = train_leg(cfg, dev, rank, world, steps=args.train_leg_steps, warmup=5, batch=4)
        except Exception as e:  # noqa: BLE001  (the headline line must survive a failure of the extra leg)
            result["train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample: a few B-clip oracle evaluations (the per-step network cost dominates a trajectory).
        # This leg is the ONLY place the CPU oracle is touched; it gets the same weights and inputs as the GPU path.
        from oracle import salunet_oracle as orc

        ocfg = orc.SalUNetConfig(img_size=cfg.img_size, up_channel=cfg.up_channel, ori_embed_dim=cfg.ori_embed_dim,
                                 down_embed_dim=cfg.down_embed_dim, num_heads=cfg.num_heads, kernel_kv=cfg.kernel_kv,
                                 temporal_list=cfg.temporal_list, dilation=cfg.dilation)
        nthreads = max(1, min(args.cpu_threads, len(os.sched_getaffinity(0))))
        torch.set_num_threads(nthreads)
        xc, fc = x_T.cpu(), [f.cpu() for f in feats]
        ac = audio.cpu() if av else None
        tcpu = torch.full((B,), 500.0)
        with torch.no_grad():
            orc.salunet_forward(sd, ocfg, xc[:1], tcpu[:1], [f[:1] for f in fc], None if ac is None else ac[:1])
            c0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                orc.salunet_forward(sd, ocfg, xc, tcpu, fc, ac)
            cdt = time.perf_counter() - c0
        result["cpu_baseline"] = {
            "value": round(B * args.cpu_steps / cdt, 4), "unit": "denoise-steps/s", "cores": nthreads, "kind": "port",
            "sample": f"{args.cpu_steps} SalUNet evaluations at batch {B} (fp32, eval, torch CPU oracle, "
                      f"{nthreads} of {os.cpu_count()} host threads: more threads are slower), "
                      f"{cdt:.1f} s; solver update excluded (negligible).  Also (BASELINE.md section 3): `batch1` = the same at one clip "
                      f"per evaluation, `dpm50_batch1` = ONE whole 50-NFE DPM-Solver trajectory of one clip (the oracle network under the "
                      f"sampler's host arithmetic)",
        }
        # BASELINE.md section 3's other two CPU points: single-clip evaluations, and one complete 50-NFE trajectory
        n1 = max(1, min(16, nthreads))
        torch.set_num_threads(n1)
        x1, f1, a1 = xc[:1], [f[:1] for f in fc], None if ac is None else ac[:1]
        with torch.no_grad():
            c0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                orc.salunet_forward(sd, ocfg, x1, tcpu[:1], f1, a1)
            c1 = time.perf_counter() - c0
            result["cpu_baseline"]["batch1"] = {"value": round(args.cpu_steps / c1, 4), "unit": "denoise-steps/s", "cores": n1,
                                                "sample": f"{args.cpu_steps} evaluations at batch 1, {c1:.1f} s"}
            if not args.no_cpu_trajectory:
                nfe = [0]

                def cpu_net(x, t, f, a=None):
                    nfe[0] += 1
                    return orc.salunet_forward(sd, ocfg, x, t.to(torch.float32), f, a)

                cpu_top = type("CpuTop", (), {"decoder_net": staticmethod(cpu_net)})()
                cpu_sampler = DiffusionSampler(cpu_top, timesteps=NFE_PER_TRAJECTORY, sample_type="dpmsolver", skip_type="logSNR",
                                               denoise=True, training_target="x0")
                c0 = time.perf_counter()
                cpu_sampler.sample_dpm_solver(x1, f1, a1)
                c2 = time.perf_counter() - c0
                result["cpu_baseline"]["dpm50_batch1"] = {
                    "value": round(nfe[0] / c2, 4), "unit": "denoise-steps/s", "cores": n1, "nfe": nfe[0], "seconds": round(c2, 2),
                    "sample": "one 50-NFE DPM-Solver trajectory (multistep-2, logSNR, denoise-to-zero) of one clip, host loop included"}
    if rank == 0:
        emit(result)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
