"""Backward kernels (training step): every autograd Function against PyTorch autograd of the
same fp32 op on the CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import salunet_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(name, *shape, scale=1.0):
    return orc.synth_tensor(name, shape, scale)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def rel(got, ref):
    ref = ref.detach().float().cpu()
    return (got.detach().float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


@pytest.fixture(scope="module")
def ag():
    from diff_sal_amd import autograd_ops, ops

    return autograd_ops, ops


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, asym, act
    (2, 14, 24, 96, 192, 3, 1, 1, 1, False, 0),
    (3, 12, 20, 64, 96, 3, 1, 2, 2, False, 1),     # dilated + ReLU
    (2, 15, 23, 96, 96, 3, 2, 0, 1, True, 0),      # Downsample (asym pad, stride 2)
    (1, 30, 46, 96, 96, 3, 4, 0, 1, True, 0),      # Downsample4x4
    (2, 10, 12, 128, 64, 1, 1, 0, 1, False, 0),    # 1x1
]


@pytest.mark.parametrize("case", CASES)
def test_conv_backward_matches_autograd(ag, case):
    agops, ops = ag
    N, H, W, Cin, Cout, k, s, p, d, asym, act = case
    x = rnd("tx%d" % Cin, N, Cin, H, W).requires_grad_(True)
    w = rnd("tw%d%d" % (Cin, Cout), Cout, Cin, k, k, scale=1.0 / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = rnd("tb", Cout, scale=0.1).requires_grad_(True)
    rv = rnd("trv", N, Cout, scale=0.3).requires_grad_(True)
    xin = F.pad(x, (0, 1, 0, 1)) if asym else x
    y = F.conv2d(xin, w, b, stride=s, padding=0 if asym else p, dilation=d) + rv[:, :, None, None]
    y = F.relu(y) if act else y
    gy = rnd("tgy", *y.shape)
    y.backward(gy)

    xd = nhwc(x.detach()).to(DEV).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    bd, rvd = b.detach().to(DEV).requires_grad_(True), rv.detach().to(DEV).requires_grad_(True)
    kw = dict(stride=(s, s), pad=(0, 0) if asym else (p, p), dil=(d, d), out_hw=tuple(y.shape[-2:]))
    yd = agops.conv(xd, ops.pack_conv_weight_diff(wd), kh=k, kw=k, bias=bd, rowvec=rvd, act=act,
                    w_dgrad=agops.dgrad_weight(wd), **kw)
    assert rel(yd, nhwc(y)) < 2e-5
    yd.backward(nhwc(gy).to(DEV))
    assert rel(xd.grad, nhwc(x.grad)) < 2e-5
    assert rel(wd.grad, w.grad) < 2e-5
    assert rel(bd.grad, b.grad) < 2e-5
    assert rel(rvd.grad, rv.grad) < 2e-5


def test_linear_backward_with_residual(ag):
    agops, ops = ag
    x = rnd("lx", 3, 50, 96).requires_grad_(True)
    w = rnd("lw", 192, 96, scale=0.1).requires_grad_(True)
    b = rnd("lb", 192, scale=0.1).requires_grad_(True)
    r = rnd("lr", 3, 50, 192).requires_grad_(True)
    y = F.linear(x, w, b) + r
    gy = rnd("lgy", 3, 50, 192)
    y.backward(gy)
    xd, wd, bd, rd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b, r))
    yd = agops.linear(xd, wd, bd, residual=rd)
    yd.backward(gy.to(DEV))
    for got, ref in ((xd.grad, x.grad), (wd.grad, w.grad), (bd.grad, b.grad), (rd.grad, r.grad)):
        assert rel(got, ref) < 2e-5
