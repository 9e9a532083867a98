"""The CPU restatement (oracle/) must reproduce the reference's golden vectors.

The vectors were produced by the real reference (oracle/gen_golden.py); this
test needs neither /root/reference nor a GPU.
"""
import numpy as np
import pytest
import torch

from oracle import salunet_oracle as orc
from tests._cases import CASES, check_taps, load_case


def _checksum(sd):
    return sum(float(sd[k].double().abs().sum()) for k in sorted(sd) if sd[k].dtype.is_floating_point)


@pytest.mark.parametrize("name", ["tiny_av", "tiny_vis", "small_av", "small_vis"])
def test_restatement_matches_reference(golden_dir, name):
    cfg, sd, x, t, feats, audio, g = load_case(golden_dir, name)
    # closed-form generator must not have drifted from the one that made the fixtures
    assert abs(_checksum(sd) - float(g["weights_checksum"])) < 1e-6 * float(g["weights_checksum"])
    taps = {}
    with torch.no_grad():
        out = orc.salunet_forward(sd, cfg, x, t, feats, audio, taps=taps)
    ref = torch.from_numpy(g["output"])
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-5
    check_taps(taps, g, 2e-5)


@pytest.mark.parametrize("name", ["full_av_b1"])
def test_restatement_full_resolution(golden_dir, name):
    cfg, sd, x, t, feats, audio, g = load_case(golden_dir, name)
    taps = {}
    with torch.no_grad():
        out = orc.salunet_forward(sd, cfg, x, t, feats, audio, taps=taps)
    assert (out - torch.from_numpy(g["output"])).abs().max().item() < 2e-5
    check_taps(taps, g, 2e-5)


def test_inputs_not_mutated(golden_dir):
    cfg, sd, x, t, feats, audio, g = load_case(golden_dir, "tiny_av")
    before = [f.clone() for f in feats]
    with torch.no_grad():
        orc.salunet_forward(sd, cfg, x, t, feats, audio)
    assert len(feats) == 4 and all(torch.equal(a, b) for a, b in zip(before, feats))


def test_f1_visual_only_output_ignores_x_and_t(golden_dir):
    """ReduceTemp (k=s=5 on T=9) only sees frames 0..4, the noise frame is #8."""
    cfg, sd, x, t, feats, audio, g = load_case(golden_dir, "tiny_vis")
    with torch.no_grad():
        a = orc.salunet_forward(sd, cfg, x, torch.tensor([10]), feats, None)
        b = orc.salunet_forward(sd, cfg, 1 - 2 * x, torch.tensor([900]), feats, None)
    assert torch.equal(a, b)


def test_state_dict_template_matches_appendix_b():
    sd = orc.state_dict_template(orc.SalUNetConfig())
    assert len(sd) == 215
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n == 37186945
    assert tuple(sd["invpt_decoder.redu_chan_up.3.proj.0.weight"].shape) == (768, 96, 5, 1, 1)
    assert tuple(sd["invpt_decoder.mid_stages.2.blocks.0.attn.conv_proj_k.conv.weight"].shape) == (192, 1, 1, 8, 8)


def test_timestep_embedding_fractional():
    e = orc.timestep_embedding(torch.tensor([998.996, 0.0]), 96)
    assert e.shape == (2, 96)
    assert torch.allclose(e[1, :48], torch.zeros(48)) and torch.allclose(e[1, 48:], torch.ones(48))
    assert abs(e[0, 0].item() - np.sin(998.996)) < 1e-4
