"""16-byte forms of the HBM-bound kernels on 16-bit storage (BASELINE configs[1] "bf16", configs[4] "fp16").

Each kernel here replaces a 4-channel (8-byte) form at the decoder's shapes.  Two checks per kernel: against the 8-byte form it
replaces (``DIFFSAL_NO_STREAM16=1`` routes a call back to it) -- bit for bit where the arithmetic order is unchanged -- and
against an fp32 PyTorch evaluation of the same 16-bit-rounded inputs within the per-operator bar of tests/test_gpu_lowp.py.
"""
import pytest
import torch

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}
OP_RTOL = {"bf16": 6e-3, "fp16": 8e-4}


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def rel_err(got, ref):
    ref = ref.float().cpu()
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


@pytest.fixture(scope="module")
def ops():
    from diff_sal_amd import ops as o

    assert torch.cuda.is_available()
    return o


class old_forms:
    """with old_forms(): the calls inside take the 8-byte forms"""

    def __enter__(self):
        from diff_sal_amd import _lib

        _lib.set_tuning("DIFFSAL_NO_STREAM16", 1)

    def __exit__(self, *a):
        from diff_sal_amd import _lib

        _lib.set_tuning("DIFFSAL_NO_STREAM16", None)


# the decoder's four stages at 224 x 384 (H, W, C, up-sampling factor of the 7 x 12 audio map), and two odd ones
FUSE_CASES = [(7, 12, 768, 7, 12), (14, 24, 384, 7, 12), (28, 48, 192, 7, 12), (56, 96, 96, 7, 12), (8, 16, 64, 2, 4), (6, 8, 128, 3, 4)]


@pytest.mark.parametrize("dname", list(DTYPES))
@pytest.mark.parametrize("H,W,C,h,w", FUSE_CASES)
def test_audio_fuse_16_byte_form(ops, dname, H, W, C, h, w):
    """K7 (R/models/saliency_decoder/transformer.py:133-146): same arithmetic order as the 8-byte form -> identical bits; the
    audio rows sit inside a wider row (the stages' align products side by side) as in the shipped step."""
    dt = DTYPES[dname]
    B, T = 2, 9
    ld = C + 96
    a_wide = rnd("s16a", B * T, h * w, ld).to(DEV).to(dt)
    a_small = a_wide[:, :, 64:64 + C]            # a 16-byte aligned channel slice of the wide rows
    x = rnd("s16x", B, T, H, W, C).to(DEV).to(dt)
    got = ops.audio_fuse(a_small, x, h, w)
    with old_forms():
        old = ops.audio_fuse(a_small, x, h, w)
    assert torch.equal(got, old)
    up = H // h
    a_up = a_small.float().view(B, T, h, w, C).repeat_interleave(up, 2).repeat_interleave(up, 3)
    m = torch.softmax((a_up * x.float()).mean(1), dim=2)
    ref = (a_up * m[:, None]).permute(0, 4, 1, 2, 3)
    assert rel_err(got, ref) < OP_RTOL[dname]
