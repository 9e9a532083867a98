"""Full-resolution pins produced by the real reference (oracle/gen_golden.py::gen_full_size): configs[1]'s real shape
(visual-only, B = 4, 224x384) with every intermediate tap, and a 50-NFE DPM-Solver trajectory at 224x384 in audio-visual mode
with strided samples of the solver state along the way.  fp32 at north_star's 1e-3; bf16 / fp16 storage at their own table
(tests/test_gpu_lowp.py: LOWP_ATOL absolute on (0,1) outputs, OP_RTOL-scaled bars on the taps)."""
import numpy as np
import pytest
import torch

from oracle import salunet_oracle as orc
from tests._cases import check_taps
from tests.test_gpu_lowp import DTYPES, LOWP_ATOL
from tests.test_gpu_lowp import build as build_lowp
from tests.test_gpu_salunet import build

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL = 1e-3
# taps on 16-bit storage: rounding of every stored activation accumulates through up to ~40 operators; measured maxima are a
# third of these bars (printed), which are relative to each tap's max |.|
TAP_RTOL = {"bf16": 6e-2, "fp16": 8e-3}


def _b4(golden_dir):
    cfg = orc.SalUNetConfig()
    g = np.load(f"{golden_dir}/salunet_full_vis_b4.npz")
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    x, feats, _ = orc.synth_inputs(cfg, 4, False, tag="full_vis_b4")
    return cfg, sd, x, torch.from_numpy(g["t"]), feats, g


@pytest.mark.parametrize("dname", ["fp32"] + list(DTYPES))
def test_full_size_batch4_visual_taps_match_reference(golden_dir, dname):
    cfg, sd, x, t, feats, g = _b4(golden_dir)
    net = build(cfg, sd) if dname == "fp32" else build_lowp(cfg, sd, DTYPES[dname])
    taps = {}
    with torch.no_grad():
        out = net(x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], None, taps=taps)
        out_fast = net(x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], None)          # the shipped (restructured) path
    st = int(g["output_stride"])
    ref = torch.from_numpy(g["output"])
    tol = RTOL * ref.abs().max().item() if dname == "fp32" else LOWP_ATOL[dname]
    for o in (out, out_fast):
        err = (o.float().cpu()[:, :, ::st, ::st] - ref).abs().max().item()
        print(f"full_vis_b4 {dname}: output max abs err {err:.3e}")
        assert err < tol
    ref_taps = {k: net.tap_to_reference_layout(k, v.float()) for k, v in taps.items()}
    worst = check_taps(ref_taps, g, RTOL if dname == "fp32" else TAP_RTOL[dname])
    print(f"full_vis_b4 {dname}: taps", {k: f"{v:.2e}" for k, v in worst.items()})
    assert {"temb", "down1", "res0", "res1", "res2", "noise0", "noise1", "noise2", "stage0", "stage1", "stage2", "stage3"} <= set(worst)


@pytest.mark.parametrize("dname", ["fp32"] + list(DTYPES))
def test_dpm_solver_50_nfe_full_size_trajectory(golden_dir, dname):
    """The solver state entering network evaluations 0 / 1 / 10 / 25 / 49 and the final x against the reference sampler driving the
    reference network (R/models/dpm_solver/sampler.py:1174-1215), AV mode, 224x384, B = 1; both the plain solver loop and the
    fused step tail (forward_fused_update)."""
    from diff_sal_amd.sampling import DiffusionSampler

    cfg = orc.SalUNetConfig()
    g = np.load(f"{golden_dir}/dpm50_full_av_b1.npz")
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    net = build(cfg, sd) if dname == "fp32" else build_lowp(cfg, sd, DTYPES[dname])
    x, feats, audio = orc.synth_inputs(cfg, 1, True, tag="dpm50_full")

    class Top(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.decoder_net, self.audio_net, self.visual_net = n, None, None

    keep = (0, 1, 10, 25, 49)
    seen, calls = {}, [0]
    f0, ff0 = net.forward, net.forward_fused_update

    def note(xx, tt):
        if calls[0] in keep:
            seen[calls[0]] = (xx.detach().float().cpu().clone(), tt.detach().float().cpu().clone())
        calls[0] += 1

    def fwd(xx, tt, f, a=None, **kw):
        note(xx, tt)
        return f0(xx, tt, f, a, **kw)

    def fwd_fused(xx, tt, f, a=None, **kw):
        note(xx, tt)
        return ff0(xx, tt, f, a, **kw)

    net.forward, net.forward_fused_update = fwd, fwd_fused
    # eager loop on purpose: the spies above must see every evaluation once (one clip would otherwise be replayed from a graph)
    s = DiffusionSampler(Top(net), timesteps=50, sample_type="dpmsolver", skip_type="logSNR", denoise=True, hip_graph=False)
    out = s.sample_dpm_solver(x.to(DEV), [f.to(DEV) for f in feats], audio.to(DEV))
    assert calls[0] == int(g["nfe"]) == 50
    ref = torch.from_numpy(g["output"])
    tol = RTOL * ref.abs().max().item() if dname == "fp32" else LOWP_ATOL[dname]
    err = (out.cpu() - ref).abs().max().item()
    print(f"dpm50_full_av_b1 {dname}: final max abs err {err:.3e}")
    assert err < tol
    for k in keep:
        xx, tt = seen[k]
        assert abs(float(tt.reshape(-1)[0]) - float(g[f"x{k}.t_input"].reshape(-1)[0])) < 1e-2
        r = torch.from_numpy(g[f"x{k}.sample"])
        e = (xx.reshape(-1)[::21] - r).abs().max().item()
        # the state is x_t = alpha x0 + sigma eps with |x| up to ~4.5: same relative bar on its own scale
        bar = (RTOL if dname == "fp32" else LOWP_ATOL[dname]) * max(1.0, float(g[f"x{k}.stats"][2]))
        print(f"  state at evaluation {k}: max abs err {e:.3e} (|x| max {float(g[f'x{k}.stats'][2]):.2f})")
        assert e < bar


def test_winograd_layers_are_active_at_batch4_and_agree_with_the_direct_kernels(golden_dir):
    """configs[1]'s real shape: the layers the library's planner moves to Winograd -- F(4x4, 3x3) on the six ResnetBlock convolutions,
    the source-resolution UpEmbed convolutions and UpEmbed-2 of stages 1 - 3 (F(2x2) only where DIFFSAL_NO_WINOGRAD4 asks for it)
    -- change the output by transform rounding only, and they do run (the outputs are not bit-equal)."""
    cfg, sd, x, t, feats, g = _b4(golden_dir)
    net = build(cfg, sd)
    args = (x.to(DEV), t.to(DEV), [f.to(DEV) for f in feats], None)
    with torch.no_grad():
        on = net(*args)
        net.winograd = False
        off = net(*args)
        net.winograd = True
    d = (on - off).abs().max().item()
    print(f"winograd on vs off at B=4: max abs diff {d:.2e}")
    assert 0.0 < d < 2e-5
