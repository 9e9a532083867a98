"""Host-side sampler logic against vectors produced by the reference sampler (tests/golden/sampler.npz)."""
import numpy as np
import pytest
import torch

from diff_sal_amd.diffusion_utils import get_beta_schedule, to_torch
from diff_sal_amd.dpm_solver import DPM_Solver, NoiseScheduleVP, interpolate_fn, model_wrapper
from oracle import salunet_oracle as orc


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/sampler.npz")


@pytest.fixture(scope="module")
def ns(gold):
    return NoiseScheduleVP("discrete", betas=to_torch(gold["betas"]))


def test_cosine_betas_match_reference(gold):
    b = get_beta_schedule("cosine", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)
    assert np.array_equal(b, gold["betas"])
    assert abs(b[0] - 4.1284e-5) < 1e-8 and b[-1] == 0.999
    for kind in ("linear", "quad", "const", "jsd", "sigmoid"):
        assert get_beta_schedule(kind, beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=10).shape == (10,)


def test_noise_schedule_tables(gold, ns):
    assert ns.total_N == int(gold["total_N"]) == 996
    t = torch.from_numpy(gold["t_grid"])
    assert np.allclose(ns.marginal_log_mean_coeff(t).numpy(), gold["log_alpha"], rtol=1e-6, atol=1e-7)
    assert np.allclose(ns.marginal_std(t).numpy(), gold["std"], rtol=1e-5, atol=1e-7)
    assert np.allclose(ns.marginal_lambda(t).numpy(), gold["lam"], rtol=1e-5, atol=1e-6)
    assert np.allclose(ns.inverse_lambda(torch.from_numpy(gold["lam"])).numpy(), gold["inv_lam"], rtol=1e-5, atol=1e-6)


def test_interpolate_fn_extrapolates_with_outer_segments():
    xp, yp = torch.tensor([[0.0, 1.0, 3.0]]), torch.tensor([[0.0, 2.0, 0.0]])
    x = torch.tensor([[-1.0], [0.5], [2.0], [4.0], [1.0]])
    assert torch.allclose(interpolate_fn(x, xp, yp).reshape(-1), torch.tensor([-2.0, 1.0, 1.0, -1.0, 2.0]))


def toy(x, t_in, img, **kw):
    return torch.sigmoid(0.7 * x + 0.001 * t_in.view(-1, 1, 1, 1) + img[0])


@pytest.mark.parametrize("algo", ["dpmsolver", "dpmsolver++"])
@pytest.mark.parametrize("skip", ["logSNR", "time_uniform"])
def test_multistep_trajectory_matches_reference(gold, ns, algo, skip):
    fn = model_wrapper(toy, ns, model_type="x_start", model_kwargs={}, guidance_type="uncond")
    solver = DPM_Solver(fn, ns, algorithm_type=algo)
    ts = solver.get_time_steps(skip, 1.0, 1.0 / ns.total_N, 49, "cpu")
    assert np.allclose(ts.numpy(), gold[f"ts.{algo}.{skip}"], rtol=2e-5, atol=2e-6)
    x = orc.synth_tensor("dpm.xT", (1, 1, 8, 12))
    img = [orc.synth_tensor("dpm.img", (1, 1, 8, 12), 0.3)]
    xe, inter = solver.sample(x, img, steps=49, order=2, skip_type=skip, method="multistep", lower_order_final=False,
                              denoise_to_zero=True, solver_type="dpmsolver", return_intermediate=True)
    assert len(inter) == 51  # 49 steps + initial + denoise-to-zero = 50 network evaluations (F2)
    ref_inter = gold[f"inter.{algo}.{skip}"]
    got = torch.stack(inter)[[0, 1, 2, 10, 25, 48, 49, 50]].numpy()
    assert np.abs(got - ref_inter).max() < 2e-4 * np.abs(ref_inter).max()
    assert np.abs(xe.numpy() - gold[f"x0.{algo}.{skip}"]).max() < 1e-4


def test_nfe_count_is_50_for_49_steps(ns):
    calls = []

    def counting(x, t_in, img, **kw):
        calls.append(float(t_in[0]))
        return torch.zeros_like(x)

    fn = model_wrapper(counting, ns, model_type="x_start")
    DPM_Solver(fn, ns, algorithm_type="dpmsolver").sample(
        torch.zeros(2, 1, 4, 4), None, steps=49, order=2, skip_type="logSNR", method="multistep",
        lower_order_final=False, denoise_to_zero=True)
    assert len(calls) == 50
    assert abs(calls[0] - 998.996) < 1e-2 and calls[-1] == 0.0  # fractional float timesteps (F3)


def test_few_steps_lower_order_final_noise_model(gold, ns):
    fn = model_wrapper(lambda x, t, img, **kw: torch.tanh(0.5 * x + img[0]), ns, model_type="noise")
    solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++")
    x = orc.synth_tensor("dpm.xT", (2, 1, 8, 12))
    img = [orc.synth_tensor("dpm.img", (2, 1, 8, 12), 0.3)]
    out = solver.sample(x, img, steps=6, order=2, skip_type="time_uniform", method="multistep",
                        lower_order_final=True, denoise_to_zero=False)
    assert np.abs(out.numpy() - gold["x0.few"]).max() < 1e-4


def test_x_start_wrapper_broadcasts_over_batch(ns):
    """Reference defect D6: its x_start branch only broadcasts for batch 1; ours handles per-sample t."""
    fn = model_wrapper(lambda x, t, img, **kw: 0.5 * x, ns, model_type="x_start")
    x = torch.randn(3, 1, 4, 5)
    t = torch.tensor([0.9, 0.5, 0.1])
    out = fn(x, t, None)
    a, s = ns.marginal_alpha(t), ns.marginal_std(t)
    ref = (x - a.view(-1, 1, 1, 1) * 0.5 * x) / s.view(-1, 1, 1, 1)
    assert torch.allclose(out, ref, atol=1e-6)


def test_unbuilt_methods_raise(ns):
    solver = DPM_Solver(lambda x, t, img=None: x, ns)
    with pytest.raises(NotImplementedError):
        solver.sample(torch.zeros(1, 1, 2, 2), steps=4, method="singlestep")


def test_diffusion_sampler_tables_match_trainer(golden_dir):
    from diff_sal_amd.sampling import DiffusionSampler

    g = np.load(f"{golden_dir}/ddim_tiny_av.npz")
    s = DiffusionSampler(model=type("M", (), {"decoder_net": None})())
    assert np.allclose(s.alphas_hat.numpy(), g["alphas_hat"], rtol=1e-6)
    assert np.allclose(s.sqrt_recip_alphas_hat.numpy(), g["sqrt_recip"], rtol=1e-6)
    assert np.allclose(s.sqrt_recipm1_alphas_hat.numpy(), g["sqrt_recipm1"], rtol=1e-5)
    assert abs(float(s.alphas_hat[500]) - 0.49229) < 1e-4


def test_ddim_loop_host_logic_with_toy_net():
    """sample_ddim on CPU tensors with a closed-form x0-predictor: compare with the textbook recursion."""
    from diff_sal_amd.sampling import DiffusionSampler

    net = lambda x, t, img, a: torch.sigmoid(x + 0.001 * t.float().view(-1, 1, 1, 1))  # noqa: E731
    s = DiffusionSampler(model=type("M", (), {"decoder_net": staticmethod(net)})(), timesteps=10)
    x = orc.synth_tensor("ddim.toy", (2, 1, 4, 6))
    out = s.sample_ddim(x.clone(), None, None)
    ah = s.alphas_hat
    ref = x.clone()
    seq = list(range(0, 1000, 100))
    for time, nxt in zip(reversed(seq), reversed([-1] + seq[:-1])):
        x0 = net(ref, torch.full((2,), time), None, None)
        if nxt < 0:
            ref = x0
            break
        eps = (torch.sqrt(1 / ah[time]) * ref - x0) / torch.sqrt(1 / ah[time] - 1)
        ref = torch.sqrt(ah[nxt]) * x0 + torch.sqrt(1 - ah[nxt]) * eps
    assert torch.allclose(out, ref, atol=1e-5)


def test_ddpm_posterior_tables_match_trainer(golden_dir):
    """posterior_* tables of R/diffusion_trainer.py:66-75 (including its sqrt(alphas_hat[t]) quirk in coef1)."""
    from diff_sal_amd.sampling import DiffusionSampler

    g = np.load(f"{golden_dir}/ddpm_tiny_av.npz")
    s = DiffusionSampler(model=type("M", (), {"decoder_net": None})())
    for k in ("posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.allclose(getattr(s, k).numpy(), g[k], rtol=1e-6, atol=1e-30), k


def test_ddpm_loop_and_legacy_ddpm_steps_host_logic():
    """sample_ddpm / p_sample on CPU tensors with a closed-form x0-predictor vs the recursion written out; the legacy
    ddpm_steps signature (noise-predicting model(x, t)) runs and returns the two lists of the reference."""
    from diff_sal_amd.sampling import DiffusionSampler, ddpm_steps

    net = lambda x, t, img, a=None: torch.sigmoid(0.5 * x + 0.001 * t.float().view(-1, 1, 1, 1))  # noqa: E731
    s = DiffusionSampler(model=type("M", (), {"decoder_net": staticmethod(net)})(), timesteps=10, sample_type="ddpm")
    x = orc.synth_tensor("ddpm.toy", (2, 1, 4, 6))
    seq = list(range(0, 1000, 100))
    zs = [orc.synth_tensor(f"ddpm.toy.z{i}", (2, 1, 4, 6)) for i in range(len(seq))]
    out = s.sample_ddpm(x.clone(), None, None, noises=zs)
    ref = x.clone()
    for i, t in enumerate(reversed(seq)):
        x0 = net(ref, torch.full((2,), t), None)
        mean = s.posterior_mean_coef1[t] * x0 + s.posterior_mean_coef2[t] * ref
        ref = mean if t == 0 else mean + zs[i] * torch.exp(0.5 * s.posterior_log_variance_clipped[t])
    assert torch.allclose(out, ref, atol=1e-6)
    assert s.sample_image(x.clone()).shape == x.shape          # dispatcher routes sample_type="ddpm"
    xs, x0s = ddpm_steps(x, seq, lambda xt, t: 0.1 * xt, s.betas)
    assert len(xs) == len(seq) + 1 and len(x0s) == len(seq) and xs[-1].shape == x.shape
    assert float(x0s[0].abs().max()) <= 1.0                      # the reference clamps x0 to [-1, 1]


def test_legacy_denoising_loops_match_the_reference(golden_dir):
    """generalized_steps / ddpm_steps / compute_alpha (call surface of R/util/denoising.py:3-69) against trajectories of
    the reference's own functions on a toy noise-predicting model (oracle/gen_golden.py::gen_legacy_denoising); the
    Gaussian draws are replayed by seeding torch the same way."""
    import numpy as np

    from diff_sal_amd.diffusion_utils import get_beta_schedule, to_torch
    from diff_sal_amd.sampling import compute_alpha, ddpm_steps, generalized_steps
    from oracle import salunet_oracle as orc

    g = np.load(f"{golden_dir}/legacy_denoising.npz")
    betas = to_torch(get_beta_schedule("cosine", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000))
    x = orc.synth_tensor("legacy.x", (2, 1, 8, 12))
    img = orc.synth_tensor("legacy.img", (2, 1, 8, 12), 0.2)
    seq = [int(v) for v in g["seq"]]
    assert np.allclose(compute_alpha(betas, torch.tensor([-1, 0, 499, 999])).numpy(), g["alpha"], rtol=1e-5, atol=1e-9)
    for eta in (0.0, 0.5):
        torch.manual_seed(77)
        if eta == 0.0:
            torch.randn(1)        # ours draws nothing at eta = 0; keep the test honest about not depending on it
        xs, x0s = generalized_steps(x, seq, lambda data, t: torch.tanh(0.3 * data["input"] + data["img"] + 0.001 * t.view(-1, 1, 1, 1)),
                                    betas, img=img, eta=eta)
        assert len(xs) == len(seq) + 1 and len(x0s) == len(seq) and xs[0] is x
        ref_xs, ref_x0 = g[f"ddim.eta{eta}.xs"], g[f"ddim.eta{eta}.x0"]
        assert np.abs(torch.stack(xs).numpy() - ref_xs).max() < 1e-4 * np.abs(ref_xs).max()
        assert np.abs(torch.stack(x0s).numpy() - ref_x0).max() < 1e-4 * np.abs(ref_x0).max()
    torch.manual_seed(78)
    xs, x0s = ddpm_steps(x, seq, lambda xt, t: torch.tanh(0.3 * xt + 0.001 * t.view(-1, 1, 1, 1)), betas)
    assert np.abs(torch.stack(xs).numpy() - g["ddpm.xs"]).max() < 1e-4 * np.abs(g["ddpm.xs"]).max()
    assert np.abs(torch.stack(x0s).numpy() - g["ddpm.x0"]).max() < 1e-4 * np.abs(g["ddpm.x0"]).max()


def test_third_order_coefficients_equal_the_finite_difference_form():
    """DPM-Solver-3 multistep (sampler.py:855-905 surface): the expanded weights of (m0, m1, m2) reproduce the published
    update written with the differences D1, D2, for both algorithm types."""
    from diff_sal_amd.diffusion_utils import get_beta_schedule, to_torch
    from diff_sal_amd.dpm_solver import DPM_Solver, NoiseScheduleVP

    ns = NoiseScheduleVP("discrete", betas=to_torch(get_beta_schedule("cosine", beta_start=1e-4, beta_end=0.02,
                                                                      num_diffusion_timesteps=1000)))
    g = torch.Generator().manual_seed(3)
    x, m0, m1, m2 = (torch.randn((2, 1, 4, 6), generator=g, dtype=torch.float64) for _ in range(4))
    ts = [torch.tensor([v]) for v in (0.9, 0.8, 0.72)]
    t = torch.tensor([0.61])
    for algo in ("dpmsolver", "dpmsolver++"):
        sol = DPM_Solver(lambda x, t, img=None: x, ns, algorithm_type=algo)
        got = sol.multistep_dpm_solver_third_update(x, [m2, m1, m0], ts, t)
        (l2, _, _), (l1, _, _), (l0, la0, s0), (lt, lat, st) = (sol._sched(v) for v in ts + [t])
        h1, h0, h = float(l1 - l2), float(l0 - l1), float(lt - l0)
        r0, r1 = h0 / h, h1 / h
        D1_0, D1_1 = (m0 - m1) / r0, (m1 - m2) / r1
        D1 = D1_0 + r0 / (r0 + r1) * (D1_0 - D1_1)
        D2 = (D1_0 - D1_1) / (r0 + r1)
        import math as _m
        if algo == "dpmsolver++":
            p1 = _m.expm1(-h); p2 = p1 / h + 1.0; p3 = p2 / h - 0.5
            a = _m.exp(float(lat))
            want = float(st / s0) * x - a * p1 * m0 + a * p2 * D1 - a * p3 * D2
        else:
            p1 = _m.expm1(h); p2 = p1 / h - 1.0; p3 = p2 / h - 0.5
            want = _m.exp(float(lat - la0)) * x - float(st) * p1 * m0 - float(st) * p2 * D1 - float(st) * p3 * D2
        assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item()


def test_fused_plan_is_the_solver_plan_and_is_cached_per_hyper_parameters():
    """DiffusionSampler._fused_plan: the per-step host scalars of the fused DPM-Solver trajectory come from DPM_Solver.plan and
    model_wrapper's time / x0 -> noise arithmetic; one table per hyper-parameter set, rebuilt when any of them changes."""
    from diff_sal_amd.sampling import DiffusionSampler

    class Top:
        decoder_net = None

    smp = DiffusionSampler(Top(), timesteps=50, sample_type="dpmsolver", skip_type="logSNR", denoise=True, training_target="x0")
    rows, last = smp._fused_plan(49)
    assert smp._fused_plan(49) is smp._plan_cache[next(iter(smp._plan_cache))]          # second call: the cached object
    ns = NoiseScheduleVP(schedule="discrete", betas=smp.betas)
    times, table, t_0 = DPM_Solver(lambda *a, **k: None, ns, algorithm_type="dpmsolver").plan(49, 2, "logSNR", None, None, False,
                                                                                          "dpmsolver")
    assert len(rows) == 49
    for s, (t_net, ex, e0, A, c0, c1, two) in enumerate(rows):
        tc = times[s].reshape(1)
        alpha, sigma = float(ns.marginal_alpha(tc)), float(ns.marginal_std(tc))
        assert t_net == float((tc - 1.0 / ns.total_N) * 1000.0)
        assert ex == 1.0 / sigma and e0 == -alpha / sigma
        assert A == table[s][0] and c0 == table[s][1][0] and two == (len(table[s][1]) > 1)
        assert c1 == (table[s][1][1] if two else 0.0)
    assert rows[0][6] is False and all(r[6] for r in rows[1:])                         # first step is first order
    assert last[0] == float((torch.ones(1) * t_0 - 1.0 / ns.total_N) * 1000.0)
    smp.timesteps = 20                                                                   # a changed hyper-parameter: new table
    rows20, _ = smp._fused_plan(19)
    assert len(rows20) == 19 and len(smp._plan_cache) == 2
    smp.skip_type = "time_uniform"
    assert smp._fused_plan(19)[0] != rows20 and len(smp._plan_cache) == 3
