import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import autograd_ops as ag
from oracle import salunet_oracle as orc
def rel(a, b): return ((a.cpu().double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()
# attention with sharp softmax
for scale_k in (1.0, 5.0, 20.0):
    n, Lq, Lk, C, heads = 3, 200, 18, 96, 2
    q, k, v = (orc.synth_tensor(nm, (n, L, C)).double().requires_grad_(True) for nm, L in (("q", Lq), ("k", Lk), ("v", Lk)))
    d = C // heads
    kk = k * scale_k
    qh, kh, vh = (t.reshape(n, -1, heads, d).transpose(1, 2) for t in (q, kk, v))
    o = (F.softmax(qh @ kh.transpose(-1, -2) * C ** -0.5, -1) @ vh).transpose(1, 2).reshape(n, Lq, C)
    go = orc.synth_tensor("go", (n, Lq, C)).double()
    o.backward(go)
    qd, kd, vd = (t.detach().float().cuda().requires_grad_(True) for t in (q, k, v))
    od = ag.attention(qd, kd * scale_k, vd, heads, C ** -0.5)
    od.backward(go.float().cuda())
    print("attention scale_k", scale_k, "fwd", rel(od.detach(), o.detach()), "dq", rel(qd.grad, q.grad), "dk", rel(kd.grad, k.grad), "dv", rel(vd.grad, v.grad))
# audio fuse with larger magnitudes
for mag in (1.0, 4.0, 10.0):
    B, T, C, ha, wa, st = 2, 9, 64, 2, 4, 3
    H, W = ha * 2 ** st, wa * 2 ** st
    x5 = (orc.synth_tensor("abx", (B, C, T, H, W)) * mag).double().requires_grad_(True)
    a_small = (orc.synth_tensor("aba", (B * T, ha * wa, C)) * mag).double().requires_grad_(True)
    a = a_small.reshape(B, T, ha, wa, C).permute(0, 4, 1, 2, 3)
    a = F.interpolate(a.reshape(B, C * T, ha, wa), scale_factor=H // ha, mode="nearest").reshape(B, C, T, H, W)
    m = F.softmax((a * x5).mean(dim=2, keepdim=True), dim=-1)
    out = a * m
    g = orc.synth_tensor("abg", tuple(out.shape)).double()
    out.backward(g)
    xd = x5.detach().float().permute(0, 2, 3, 4, 1).contiguous().cuda().requires_grad_(True)
    ad = a_small.detach().float().cuda().requires_grad_(True)
    od = ag.audio_fuse(ad, xd, ha, wa)
    od.backward(g.float().cuda())
    print("audio_fuse mag", mag, "fwd", rel(od.detach(), out.detach()), "dx", rel(xd.grad, x5.grad.permute(0, 2, 3, 4, 1)), "da", rel(ad.grad, a_small.grad))
