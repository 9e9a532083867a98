"""Sampling loops on the GPU against trajectories produced by the reference trainer / reference DPM-Solver."""
import numpy as np
import pytest
import torch

from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Top(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.decoder_net = net
        self.audio_net = None
        self.visual_net = None


@pytest.fixture(scope="module")
def tiny():
    cfg = CASES["tiny_av"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    x, feats, audio = orc.synth_inputs(cfg, 1, True, tag="ddim")
    return Top(build(cfg, sd)), x.to(DEV), [f.to(DEV) for f in feats], audio.to(DEV)


def test_sample_ddim_10_steps_matches_reference_trainer(golden_dir, tiny):
    """Config 1 plumbing: the reference trainer's own sample_ddim, 10 steps, eta 0 (R/diffusion_trainer.py:440-480)."""
    from diff_sal_amd.sampling import DiffusionSampler

    top, x, feats, audio = tiny
    g = np.load(f"{golden_dir}/ddim_tiny_av.npz")
    out = DiffusionSampler(top, timesteps=10, sample_type="ddim").sample_ddim(x, feats, audio)
    ref = torch.from_numpy(g["output"])
    assert (out.cpu() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()
    assert len(feats) == 4 and feats[0].shape[2] == 8  # caller's list untouched, no deep copies needed


def test_dpm_solver_50_nfe_matches_reference_sampler(golden_dir, tiny):
    """49 multistep-2 steps + denoise-to-zero through the real reference sampler with the same net."""
    from diff_sal_amd.sampling import DiffusionSampler

    top, x, feats, audio = tiny
    g = np.load(f"{golden_dir}/dpm50_tiny_av.npz")
    s = DiffusionSampler(top, timesteps=50, sample_type="dpmsolver", skip_type="logSNR", denoise=True)
    out = s.sample_dpm_solver(x, feats, audio)
    ref = torch.from_numpy(g["output"])
    assert (out.cpu() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


def test_sampling_loop_has_no_host_sync(tiny):
    """The loop must be enqueue-only: run it under a stream capture-like check by timing enqueue vs completion."""
    from diff_sal_amd.sampling import DiffusionSampler

    top, x, feats, audio = tiny
    s = DiffusionSampler(top, timesteps=10, sample_type="dpmsolver")
    s.sample_dpm_solver(x, feats, audio)  # warm-up (weight packing, library load)
    torch.cuda.synchronize()
    with torch.cuda.stream(torch.cuda.Stream()):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):  # capture fails on any synchronising call or host read-back
            y = s.sample_dpm_solver(x, feats, audio)
    g.replay()
    torch.cuda.synchronize()
    eager = s.sample_dpm_solver(x, feats, audio)
    assert torch.equal(y, eager)


def test_hip_graph_mode_and_f1_shortcut_agree_with_the_plain_loop():
    from diff_sal_amd.sampling import DiffusionSampler

    cfg = CASES["tiny_vis"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    top = Top(build(cfg, sd))
    x, feats, _ = orc.synth_inputs(cfg, 2, False, tag="modes")
    xd, fd = x.to(DEV), [f.to(DEV) for f in feats]
    plain = DiffusionSampler(top, timesteps=8, sample_type="dpmsolver").sample_dpm_solver(xd, fd, None)
    graphed = DiffusionSampler(top, timesteps=8, sample_type="dpmsolver", hip_graph=True)
    g1 = graphed.sample_dpm_solver(xd, fd, None)
    g2 = graphed.sample_dpm_solver(xd * 0.5, fd, None)  # replay with new inputs
    assert torch.equal(g1, plain)
    assert torch.isfinite(g2).all()
    # F1: visual-only output ignores (x, t): one evaluation reproduces the 8-step sample up to fp32 rounding
    short = DiffusionSampler(top, timesteps=8, sample_type="dpmsolver", step_invariant_shortcut=True)
    assert (short.sample_dpm_solver(xd, fd, None) - plain).abs().max().item() < 1e-5
    ddim = DiffusionSampler(top, timesteps=8, sample_type="ddim")
    assert (short.sample_ddim(xd, fd, None) - ddim.sample_ddim(xd, fd, None)).abs().max().item() < 1e-6


def test_sample_ddpm_10_steps_matches_reference_trainer(golden_dir):
    """DDPM ancestral sampling: the reference trainer's own p_sample loop (R/diffusion_trainer.py:482-527, 574-580), 10 steps,
    visual conditioning, Gaussian draws replayed from closed-form tensors."""
    from diff_sal_amd.sampling import DiffusionSampler

    cfg = CASES["tiny_av"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    x, feats, _ = orc.synth_inputs(cfg, 1, True, tag="ddpm")
    g = np.load(f"{golden_dir}/ddpm_tiny_av.npz")
    zs = [orc.synth_tensor(f"ddpm.z{i}", tuple(x.shape)).to(DEV) for i in range(10)]
    s = DiffusionSampler(Top(build(cfg, sd)), timesteps=10, sample_type="ddpm")
    out = s.sample_ddpm(x.to(DEV), [f.to(DEV) for f in feats], None, noises=zs)
    ref = torch.from_numpy(g["output"])
    assert (out.cpu() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


def test_sampling_trajectories_in_bf16x3_mode_stay_inside_the_parity_bar(golden_dir, tiny):
    """The opt-in split-precision GEMM mode through whole trajectories: 10-step DDIM of the reference trainer and the 50-NFE
    DPM-Solver of the reference sampler.  Bar 1e-3 relative (north_star); the error does not accumulate over the steps."""
    from diff_sal_amd import ops
    from diff_sal_amd.sampling import DiffusionSampler

    top, x, feats, audio = tiny
    ops.set_gemm_precision("bf16x3")
    try:
        ddim = DiffusionSampler(top, timesteps=10, sample_type="ddim").sample_ddim(x, feats, audio)
        dpm = DiffusionSampler(top, timesteps=50, sample_type="dpmsolver").sample_dpm_solver(x, feats, audio)
    finally:
        ops.set_gemm_precision("fp32")
    for name, out in (("ddim_tiny_av", ddim), ("dpm50_tiny_av", dpm)):
        ref = torch.from_numpy(np.load(f"{golden_dir}/{name}.npz")["output"])
        err = (out.cpu() - ref).abs().max().item()
        print(name, "bf16x3 max abs err", err)
        assert err < 1e-3 * ref.abs().max().item()


def test_hip_graph_is_recaptured_after_parameters_or_hyperparameters_change():
    """A captured trajectory bakes in the packed-weight pointers and the sampler's coefficients: replay after an optimizer
    step / load_state_dict / a changed ``timesteps`` must not return the stale result."""
    from diff_sal_amd.sampling import DiffusionSampler
    from diff_sal_amd.train_step import DiffusionTrainStep

    cfg = CASES["tiny_av"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    net = build(cfg, sd)
    top = Top(net)
    x, feats, audio = orc.synth_inputs(cfg, 2, True, tag="regraph")
    xd, fd, ad = x.to(DEV), [f.to(DEV) for f in feats], audio.to(DEV)
    g = DiffusionSampler(top, timesteps=4, sample_type="ddim", hip_graph=True)
    plain = DiffusionSampler(top, timesteps=4, sample_type="ddim")
    assert torch.equal(g.sample_ddim(xd, fd, ad), plain.sample_ddim(xd, fd, ad))
    # one training step rewrites the parameters in place (fused Adam through raw pointers)
    ts = DiffusionTrainStep(net, lr=1e-2)
    net.dropout_p = 0.0
    ts.step(torch.sigmoid(xd), {"feat_list": fd, "audio_feat": ad}, t0=300)
    net.eval()
    after_plain = plain.sample_ddim(xd, fd, ad)
    after_graph = g.sample_ddim(xd, fd, ad)
    assert (after_plain - plain.sample_ddim(xd, fd, ad)).abs().max().item() == 0.0
    assert torch.equal(after_graph, after_plain)
    # sampler hyper-parameter change
    g.timesteps = plain.timesteps = 6
    assert torch.equal(g.sample_ddim(xd, fd, ad), plain.sample_ddim(xd, fd, ad))
    # precision change on the module
    net.gemm_precision = "bf16x3"
    assert torch.equal(g.sample_ddim(xd, fd, ad), plain.sample_ddim(xd, fd, ad))
    net.gemm_precision = None


def test_fused_step_tail_is_bit_equal_to_the_plain_solver_loop(golden_dir, tiny):
    """final resize + x0->noise conversion + multistep update folded into the denoiser's last kernel.
    Same arithmetic per element as the stand-alone kernels: identical trajectories, and both match the reference's."""
    from diff_sal_amd.sampling import DiffusionSampler

    top, x, feats, audio = tiny
    kw = dict(timesteps=50, sample_type="dpmsolver", skip_type="logSNR", denoise=True)
    fused = DiffusionSampler(top, fused_update=True, **kw)
    plain = DiffusionSampler(top, fused_update=False, **kw)
    assert fused._fusable(x) and not plain._fusable(x)
    a, b = fused.sample_dpm_solver(x, feats, audio), plain.sample_dpm_solver(x, feats, audio)
    assert torch.equal(a, b)
    ref = torch.from_numpy(np.load(f"{golden_dir}/dpm50_tiny_av.npz")["output"])
    assert (a.cpu() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()
    # few steps + lower_order_final + time_uniform grid + batch > 1, visual-only
    x2 = torch.cat([x, 0.5 * x])
    f2 = [torch.cat([f, f.flip(2)]) for f in feats]
    kw2 = dict(timesteps=7, sample_type="dpmsolver", skip_type="time_uniform", denoise=False, lower_order_final=True)
    assert torch.equal(DiffusionSampler(top, fused_update=True, **kw2).sample_dpm_solver(x2, f2, None),
                       DiffusionSampler(top, fused_update=False, **kw2).sample_dpm_solver(x2, f2, None))
    # the ops it replaces, directly
    from diff_sal_amd import ops
    s_low = torch.rand(2, 32, 64, 1, device=DEV)
    xs, mp = torch.randn(2, 1, 64, 128, device=DEV), torch.randn(2, 1, 64, 128, device=DEV)
    m, xn, x0 = ops.resize_update(s_low, xs, mp, 1.7, -0.6, 0.9, -0.3, 0.2, want_x0=True)
    x0_ref = ops.resize_bilinear(s_low, 64, 128).view(2, 1, 64, 128)
    m_ref = ops.axpbypcz(xs, 1.7, x0_ref, -0.6)
    assert torch.equal(x0, x0_ref) and torch.equal(m, m_ref) and torch.equal(xn, ops.axpbypcz(xs, 0.9, m_ref, -0.3, mp, 0.2))


def test_default_sampler_replays_small_batches_from_a_graph_and_matches_the_eager_loop():
    """DiffusionSampler's default (hip_graph="auto"): one or two clips are replayed from a captured trajectory, three or more run
    eagerly; both give the eager loop's result bit for bit, and repeated calls with new inputs reuse the capture."""
    from diff_sal_amd.sampling import DiffusionSampler

    cfg = CASES["tiny_av"][0]
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    top = Top(build(cfg, sd))
    auto = DiffusionSampler(top, timesteps=8, sample_type="dpmsolver")
    eager = DiffusionSampler(top, timesteps=8, sample_type="dpmsolver", hip_graph=False)
    assert auto.hip_graph == "auto" and eager.hip_graph is False
    for B in (1, 2, 3):
        for tag in ("a", "b"):
            x, feats, audio = orc.synth_inputs(cfg, B, True, tag=f"auto{B}{tag}")
            xd, fd, ad = x.to(DEV), [f.to(DEV) for f in feats], audio.to(DEV)
            assert torch.equal(auto.sample_dpm_solver(xd, fd, ad), eager.sample_dpm_solver(xd, fd, ad))
    assert len(auto._graphs) == 2 and not eager._graphs          # captured for B = 1 and B = 2 only
