"""Host-side surface checks (no GPU): checkpoint formats, train-mode batching, bench launcher, sampler guards."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from torch import nn

from tests._cases import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tiny_salunet(**extra):
    from diff_sal_amd.sal_unet import SalUNet

    cfg = CASES["tiny_vis"][0]
    return SalUNet(image_based=True, img_size=cfg.img_size, frames_len=1, mid_num_stages=4, temporal_size=9,
                   temporal_list=[5] * 4, futr_num_stages=0, ori_embed_dim=256, down_embed_dim=32,
                   idx_to_planes={0: 32, 1: 192, 2: 384, 3: 256}, patch_size=[0, 3, 3, 3], patch_stride=[0, 1, 1, 1],
                   patch_padding=[0, 2, 2, 2], up_channel=[256, 128, 64, 32], num_heads=[2] * 4, mlp_ratio=[2.0] * 4,
                   drop_path_rate=[0.15] * 4, qkv_bias=[True] * 4, kv_proj_method=["avg"] * 4, kernel_kv=[2, 4, 8, 16],
                   padding_kv=[0] * 4, stride_kv=[2, 4, 8, 16], q_proj_method=["dw_bn"] * 4, kernel_q=[3] * 4,
                   padding_q=[1] * 4, stride_q=[1] * 4, **extra)


def test_prefixed_checkpoint_loads_like_the_reference_does():
    """The reference saves ``model.state_dict()`` of the DDP/DataParallel wrapper -- keys ``module.decoder_net.<k>`` (+ the
    encoders') -- and loads it back with ``strict=0`` (R/model.py:17-22, R/diffusion_trainer.py:263-265).  The same
    checkpoint must land every one of the 215 denoiser tensors in VideoSaliencyModel(decoder_net=SalUNet)."""
    from diff_sal_amd.diff_model import VideoSaliencyModel

    src = tiny_salunet()
    g = torch.Generator().manual_seed(5)
    sd = {k: (torch.randn(v.shape, generator=g) if v.dtype.is_floating_point else torch.full_like(v, 7))
          for k, v in src.state_dict().items()}
    assert len(sd) == 215
    ckpt = {"state_dict": {f"module.decoder_net.{k}": v for k, v in sd.items()}}
    ckpt["state_dict"]["module.visual_net.patch_embed.proj.weight"] = torch.zeros(96, 3, 3, 7, 7)   # encoder keys: ignored
    ckpt["state_dict"]["module.fc.0.weight"] = torch.zeros(512, 128)

    class Wrapper(nn.Module):            # what DataParallel / DDP look like to load_state_dict
        def __init__(self, m):
            super().__init__()
            self.module = m

    model = Wrapper(VideoSaliencyModel(channel_list=None, decoder_net=dict(type="SalUNet", **_tiny_kwargs())))
    res = model.load_state_dict(ckpt["state_dict"], strict=False)
    assert not [k for k in res.missing_keys if "decoder_net" in k]
    assert sorted(res.unexpected_keys) == ["module.fc.0.weight", "module.visual_net.patch_embed.proj.weight"]
    got = model.module.decoder_net.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)


def _tiny_kwargs():
    cfg = CASES["tiny_vis"][0]
    return dict(image_based=True, img_size=cfg.img_size, frames_len=1, mid_num_stages=4, temporal_size=9,
                temporal_list=[5] * 4, futr_num_stages=0, ori_embed_dim=256, down_embed_dim=32,
                idx_to_planes={0: 32, 1: 192, 2: 384, 3: 256}, patch_size=[0, 3, 3, 3], patch_stride=[0, 1, 1, 1],
                patch_padding=[0, 2, 2, 2], up_channel=[256, 128, 64, 32], num_heads=[2] * 4, mlp_ratio=[2.0] * 4,
                drop_path_rate=[0.15] * 4, qkv_bias=[True] * 4, kv_proj_method=["avg"] * 4, kernel_kv=[2, 4, 8, 16],
                padding_kv=[0] * 4, stride_kv=[2, 4, 8, 16], q_proj_method=["dw_bn"] * 4, kernel_q=[3] * 4,
                padding_q=[1] * 4, stride_q=[1] * 4)


def test_training_batches_are_never_chunked():
    """BatchNorm batch statistics are over the whole per-rank batch (reference cfg batch_size 48); eval batches beyond the pass size
    (``clips_per_pass``: what keeps every operand inside 32-bit offsets, capped by ``max_clips_per_pass``) are evaluated in passes."""
    net = tiny_salunet().train()
    seen = []
    net.forward_train = lambda x, t, f, a=None: seen.append(x.shape[0]) or x
    x = torch.zeros(40, 1, 64, 128)
    net(x, torch.zeros(40, dtype=torch.long), [None] * 4, None)
    assert seen == [40]
    net.eval()
    calls = []
    net._forward_pass = lambda x, t, f, a, taps: calls.append(x.shape[0]) or torch.zeros(x.shape[0], 1, 64, 128)
    net(x, torch.zeros(40, dtype=torch.long), [torch.zeros(40, 1)] * 4, None)
    assert calls == [40]                       # small maps: 40 clips fit one pass
    net.max_clips_per_pass = 16
    calls.clear()
    net(x, torch.zeros(40, dtype=torch.long), [torch.zeros(40, 1)] * 4, None)
    assert calls == [16, 16, 8]


def test_optimizer_state_dict_is_the_torch_adam_format():
    """DiffusionTrainStep.state_dict()/load_state_dict() interchange with torch.optim.Adam over the same module
    (the reference's ``optim_dict``, R/diffusion_trainer.py:187-193, 263-268)."""
    from diff_sal_amd.train_step import DiffusionTrainStep

    def make():
        torch.manual_seed(0)
        m = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
        m[0].bias.requires_grad_(False)                       # an untrained parameter in the middle of the numbering
        return m

    ref_m = make()
    opt = torch.optim.Adam(ref_m.parameters(), lr=3e-4, betas=(0.8, 0.95), eps=1e-7, weight_decay=0.01)
    for _ in range(3):
        opt.zero_grad()
        ref_m(torch.randn(4, 6)).square().sum().backward()
        opt.step()
    ts = DiffusionTrainStep(make())
    ts.load_state_dict(opt.state_dict())
    assert ts.step_count == 3 and ts.lr == 3e-4 and ts.betas == (0.8, 0.95) and ts.eps == 1e-7 and ts.weight_decay == 0.01
    out = ts.state_dict()
    ref_sd = opt.state_dict()
    assert set(out["state"]) == set(ref_sd["state"]) == {0, 2, 3}
    for i in out["state"]:
        assert float(out["state"][i]["step"]) == 3.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(out["state"][i][k], ref_sd["state"][i][k])
    assert out["param_groups"][0]["params"] == [0, 1, 2, 3]
    # and the other way round: our dict loads into a fresh torch Adam
    opt2 = torch.optim.Adam(make().parameters())
    opt2.load_state_dict(out)
    assert opt2.state_dict()["param_groups"][0]["lr"] == 3e-4
    # validation: wrong module -> clear error, nothing silently misplaced
    other = DiffusionTrainStep(nn.Sequential(nn.Linear(6, 5), nn.Linear(5, 4)))
    with pytest.raises(ValueError):
        other.load_state_dict(out)
    bad = {"state": {0: {"step": torch.tensor(1.0), "exp_avg": torch.zeros(2, 2), "exp_avg_sq": torch.zeros(2, 2)}},
           "param_groups": [dict(out["param_groups"][0])]}
    with pytest.raises(ValueError, match="shape"):
        ts.load_state_dict(bad)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` starts two rank processes itself (no torchrun needed) and prints ONE JSON line with
    n_gpus = 2; on this GPU-less host the ranks rendezvous over gloo and say that no workload ran."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("plumbing-only mode is for GPU-less hosts")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["value"] is None and len(j["devices"]) == 2
    # the default line carries a `train` object at every N (BASELINE configs[3]); here its gradient-exchange plumbing ran over gloo
    tr = j["train"]
    ex = tr["exchange"]
    assert tr["n_gpus"] == 2 and ex["world"] == 2 and ex["collective_executed"] and ex["replicas_identical"] is True
    assert ex["bytes_per_step_per_rank"] > 0 and len(ex["buckets_mb"]) == tr["grad_buckets"] >= 2
    assert sorted(ex["launch_order"]) == list(range(tr["grad_buckets"]))
    for k in ("allreduce_ms_per_bucket", "ms_per_step_without_exchange", "exposed_ms", "overlap_frac"):
        assert k in ex
    for wl in ("train",):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", wl, "--steps", "1",
                            "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 2


@pytest.mark.parametrize("mode", ["allreduce", "reduce_scatter"])
def test_bench_train_plumbing_at_eight_ranks_keeps_the_replicas_identical(mode):
    """`bench.py --workload train --gpus 8` on a GPU-less host: eight gloo ranks run three optimizer steps of the exchange
    plumbing, each rank at its OWN timestep per step (the reference draws t0 per rank-batch from an unseeded generator,
    R/diffusion_trainer.py:111) -- the replicas must stay bit-identical, in either exchange form."""
    if torch.cuda.is_available():
        pytest.skip("plumbing-only mode is for GPU-less hosts")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "train", "--steps", "3", "--warmup", "1",
                        "--exchange", mode], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    ex = j["exchange"]
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and j["unit"] == "samples/s" and j["value"] is None
    assert ex["world"] == 8 and ex["steps"] == 3 and ex["rank_dependent_t0"] and ex["mode"] == mode
    assert ex["collective_executed"] and ex["replicas_identical"] is True


def test_step_invariant_shortcut_checks_its_precondition():
    from diff_sal_amd.sampling import DiffusionSampler

    class Top(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.decoder_net = n

    ok = tiny_salunet()
    DiffusionSampler(Top(ok), step_invariant_shortcut=True)                      # temporal_list 5 < 9: fine
    with pytest.raises(ValueError, match="training_target"):
        DiffusionSampler(Top(ok), step_invariant_shortcut=True, training_target="noise")
    reaches = tiny_salunet()
    reaches.temporal_list = [9, 9, 9, 9]
    with pytest.raises(ValueError, match="noise frame"):
        DiffusionSampler(Top(reaches), step_invariant_shortcut=True)
    not_ib = tiny_salunet()
    not_ib.image_based = False
    with pytest.raises(ValueError, match="image_based"):
        DiffusionSampler(Top(not_ib), step_invariant_shortcut=True)


def test_rel_table_plan_reproduces_interpolate_and_gather_and_its_transpose():
    """MViT._rel_plan (the sparse row map the training path hands to diffsal_rel_tables) against the torch form it replaces
    (F.interpolate(mode='linear') + index gather, R/models/mvit.py:330-361), forward and transposed (CSR) -- on the host."""
    from diff_sal_amd.mvit import MViT

    enc = MViT(arch="tiny", out_scales=[3])
    g = torch.Generator().manual_seed(5)
    for rel_len, q, k in ((15, 8, 8), (111, 56, 7), (111, 96, 12), (27, 14, 7), (27, 24, 12), (13, 7, 7), (13, 12, 12), (55, 28, 7)):
        rel = torch.randn(rel_len, 96, generator=g, requires_grad=True)
        ref = MViT._rel_table(rel, q, k)                               # detached torch reference
        pl = enc._rel_plan(rel_len, q, k, "cpu")
        i2, w2 = pl["idx2"].long(), pl["w2"]
        got = (w2[:, :1] * rel.detach()[i2[:, 0]] + w2[:, 1:] * rel.detach()[i2[:, 1]]).view(q, k, 96)
        assert torch.allclose(got, ref, rtol=0, atol=2e-5), (rel_len, q, k, (got - ref).abs().max())   # lambda in float32: a few ulp from ATen
        # transposed map against autograd of the torch form
        max_rel = 2 * max(q, k) - 1
        r = rel
        if rel_len != max_rel:
            r = torch.nn.functional.interpolate(r.t().unsqueeze(0), size=max_rel, mode="linear").squeeze(0).t()
        q_ratio, k_ratio = max(k / q, 1.0), max(q / k, 1.0)
        idx = ((torch.arange(q)[:, None] * q_ratio - torch.arange(k)[None, :] * k_ratio) + (k - 1) * k_ratio).long()
        dout = torch.randn(q, k, 96, generator=g)
        (r[idx] * dout).sum().backward()
        ptr, col, cw = pl["csr_ptr"].long(), pl["csr_col"].long(), pl["csr_w"]
        d = torch.zeros(rel_len, 96)
        flat = dout.view(-1, 96)
        for row in range(rel_len):
            for e in range(int(ptr[row]), int(ptr[row + 1])):
                d[row] += cw[e] * flat[col[e]]
        assert torch.allclose(d, rel.grad, rtol=1e-4, atol=1e-4), (rel_len, q, k, (d - rel.grad).abs().max())


def test_train_step_optimizer_face_drives_lr_through_torch_schedulers():
    """R/util/utils.py:116-123 wraps Adam in MultiStepLR and R/diffusion_trainer.py:296 steps it once per epoch: the same
    scheduler object must be able to drive DiffusionTrainStep's flat Adam (host logic only: no optimizer step runs here)."""
    import warnings

    import torch
    from torch import nn

    from diff_sal_amd.train_step import DiffusionTrainStep, FlatAdam

    net = nn.Sequential(nn.Linear(8, 8), nn.Linear(8, 4))
    ts = DiffusionTrainStep(net, lr=1e-4, weight_decay=0.0)
    assert isinstance(ts.optimizer, FlatAdam) and isinstance(ts.optimizer, torch.optim.Optimizer)
    assert len(ts.param_groups) == 1 and ts.param_groups[0]["lr"] == 1e-4
    assert sum(p.numel() for p in ts.param_groups[0]["params"]) == sum(p.numel() for p in net.parameters())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # "lr_scheduler.step() before optimizer.step()": no HIP step on this host
        sched = torch.optim.lr_scheduler.MultiStepLR(ts.optimizer, milestones=[2, 4], gamma=0.1)
        lrs = []
        for _ in range(5):
            lrs.append(ts.lr)
            sched.step()
    assert lrs == pytest.approx([1e-4, 1e-4, 1e-5, 1e-5, 1e-6])
    ts.lr = 3e-4                                   # the attribute and the group are one value
    assert ts.param_groups[0]["lr"] == 3e-4
    sd = ts.optimizer.state_dict()                 # torch.optim.Adam checkpoint format, from the owner
    assert sd["param_groups"][0]["lr"] == 3e-4 and sd["state"] == {}
    with pytest.raises(RuntimeError):
        ts.optimizer.add_param_group({"params": [nn.Parameter(torch.zeros(3))]})


def test_train_step_loss_hook_arguments():
    import torch
    from torch import nn

    from diff_sal_amd.train_step import DiffusionTrainStep

    net = nn.Linear(4, 4)
    with pytest.raises(ValueError):
        DiffusionTrainStep(net, loss_fn=lambda p, g: (p - g).square().sum(), loss_config=object())
    cfg = type("C", (), {"loss": type("L", (), dict(loss_kl=False, loss_ce=False, loss_mse=True, mse_weight=1.0, loss_cc=True, cc_weight=-0.1,
                                                     loss_sim=False, sim_weight=0.0, loss_nss=False, nss_weight=0.0))()})()
    ts = DiffusionTrainStep(nn.Linear(4, 4), loss_config=cfg)
    assert callable(ts.loss_fn) and ts.last_losses is None


def test_bench_ends_every_rank_when_one_dies():
    """A rank that exits before the rendezvous leaves its peers waiting in init_process_group forever: spawn_ranks' watchdog
    must notice the death, stop the others and return a failure code promptly (not after the job timeout)."""
    import time

    if torch.cuda.is_available():
        pytest.skip("plumbing-only mode is for GPU-less hosts")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DIFFSAL_BENCH_FAULT_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--no-cpu-baseline", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert "rank failure" in r.stderr
    assert time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no result line from a broken job


def test_loss_ce_main_term_matches_reference_value(golden_dir):
    """cross_entropy_loss (R/models/sal_losses.py:48-63) on the host: the main term of the loss_ce configuration against the
    value the reference produced (tests/golden/sal_loss_ce.npz); needs no HIP library (no saliency term enabled)."""
    import types

    from diff_sal_amd import sal_losses as sl

    g = np.load(f"{golden_dir}/sal_loss_ce.npz")
    lc = dict(loss_kl=False, loss_ce=True, loss_mse=False, loss_cc=False, loss_sim=False, loss_nss=False, ce_weight=float(g["cfg"][0]))
    cfg = types.SimpleNamespace(loss=types.SimpleNamespace(**lc))
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    main, cc, sim, nss = sl.get_kl_cc_sim_loss(cfg, pred, torch.from_numpy(g["gt"]))
    assert abs(float(main) - float(g["main"])) <= 1e-5 * abs(float(g["main"]))
    assert float(cc) == 0.0 and float(sim) == 0.0 and float(nss) == 0.0
    main.backward()
    assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().max()) > 0
