"""Shared definitions of the golden cases (mirrors oracle/gen_golden.py::CASES)."""
import numpy as np
import torch

from oracle import salunet_oracle as orc

TINY = dict(img_size=(64, 128), up_channel=(256, 128, 64, 32), ori_embed_dim=256, down_embed_dim=32)

CASES = {
    "small_av": (orc.SalUNetConfig(img_size=(64, 128)), 2, True),
    "small_vis": (orc.SalUNetConfig(img_size=(64, 128)), 2, False),
    "tiny_av": (orc.SalUNetConfig(**TINY), 2, True),
    "tiny_vis": (orc.SalUNetConfig(**TINY), 1, False),
    "full_av_b1": (orc.SalUNetConfig(), 1, True),
    "full_vis_b1": (orc.SalUNetConfig(), 1, False),
}


def load_case(golden_dir, name):
    cfg, batch, av = CASES[name]
    g = np.load(f"{golden_dir}/salunet_{name}.npz")
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    x, feats, audio = orc.synth_inputs(cfg, batch, av, tag=name)
    t = torch.from_numpy(g["t"])
    return cfg, sd, x, t, feats, audio, g


def check_taps(taps, g, rtol, names=None):
    """Compare strided samples of intermediate taps against the fixture (relative to tap max)."""
    worst = {}
    for key in g.files:
        if not key.endswith(".sample"):
            continue
        name = key.split(".")[1]
        if name not in taps or (names is not None and name not in names):
            continue
        got = taps[name].detach().float().cpu()
        assert tuple(got.shape) == tuple(g[f"tap.{name}.shape"]), (name, got.shape)
        stride = int(g[f"tap.{name}.stride"])
        ref = torch.from_numpy(g[key])
        err = (got.reshape(-1)[::stride] - ref).abs().max().item() / (float(g[f"tap.{name}.stats"][2]) + 1e-12)
        worst[name] = err
        assert err < rtol, (name, err)
    return worst


def train_fixture_inputs(golden_dir, name):
    """Inputs of oracle/gen_golden.py::gen_train_step (closed form) + the fixture."""
    av = name.endswith("_av")
    cfg = CASES["tiny_av"][0]
    g = np.load(f"{golden_dir}/{name}.npz")
    sd = orc.synth_state_dict(orc.state_dict_template(cfg))
    tag = str(g["tag"])
    noise, feats, audio = orc.synth_inputs(cfg, 2, av, tag=tag)
    sal = torch.sigmoid(orc.synth_tensor(tag + ".sal", (2, 1, *cfg.img_size)))
    dq = orc.synth_tensor(tag + ".dq", tuple(sal.shape))
    return cfg, sd, sal, dq, noise, int(g["t0"]), feats, audio, g


def sampled_err(got, g, key):
    """max |got - fixture| over the fixture's strided samples of tensor ``key``, and the fixture tensor's max |.|."""
    got = got.detach().float().cpu()
    assert tuple(got.shape) == tuple(g[f"tap.{key}.shape"]), (key, got.shape)
    stride = int(g[f"tap.{key}.stride"])
    ref = torch.from_numpy(g[f"tap.{key}.sample"])
    return (got.reshape(-1)[::stride] - ref).abs().max().item(), float(g[f"tap.{key}.stats"][2])
