"""d(loss)/d(activation) at the 11 ReLU outputs of the decoder: HIP autograd vs the fp64 oracle (debug aid).
usage: grad_trace.py TAG T0 [av]"""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
from diff_sal_amd import autograd_ops as ag
cfg = CASES["tiny_av"][0]
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
tag, t0 = sys.argv[1], int(sys.argv[2])
av = len(sys.argv) > 3 and sys.argv[3] == "av"
x, feats, audio = orc.synth_inputs(cfg, 2, av, tag=tag)
x0 = torch.sigmoid(orc.synth_tensor(tag + ".x0", (2, 1, *cfg.img_size)))
t = torch.tensor([t0, t0])
if "qs" in sys.argv:   # the inputs DiffusionTrainStep.prepare_data builds in tests/test_gpu_train_step.py
    from diff_sal_amd.diffusion_utils import get_beta_schedule, to_torch
    a_hat = (1.0 - to_torch(get_beta_schedule("cosine", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000))).cumprod(dim=0)
    sal = torch.sigmoid(orc.synth_tensor(tag + ".sal", (2, 1, *cfg.img_size)))
    x0 = sal + 0.01 * orc.synth_tensor(tag + ".dq", tuple(sal.shape))
    x = a_hat[t0].sqrt() * x0 + (1 - a_hat[t0]).sqrt() * x
_orig_emb = orc.timestep_embedding
orc.timestep_embedding = lambda tt, d: _orig_emb(tt, d).double()
ref = []
orig_relu = F.relu
def spy_relu(v, *a, **k):
    y = orig_relu(v, *a, **k); y.retain_grad(); ref.append(y); return y
orc.F.relu = spy_relu
orc.BN_TRAIN = True
leaf = {k: (v.double() if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
pred = orc.salunet_forward(leaf, cfg, x.double(), t, [f.double() for f in feats], None if audio is None else audio.double())
orc.BN_TRAIN = False
orc.F.relu = orig_relu
((pred - x0.double()) ** 2).sum(dim=(1, 2, 3)).mean().backward()
hip = []
o_bn, o_conv = ag.batchnorm_relu_train, ag.conv
def spy_bn(x_, bn, relu=True):
    y = o_bn(x_, bn, relu); y.retain_grad(); hip.append(("bn", y)); return y
def spy_conv(x_, w, **k):
    y = o_conv(x_, w, **k)
    if k.get("act", 0) == 1: y.retain_grad(); hip.append(("conv", y))
    return y
ag.batchnorm_relu_train, ag.conv = spy_bn, spy_conv
net = build(cfg, sd); net.train(); net.dropout_p = 0.0
out = net(x.cuda(), t.cuda(), [f.cuda() for f in feats], None if audio is None else audio.cuda())
((out - x0.cuda()) ** 2).sum(dim=(1, 2, 3)).mean().backward()
names = ["s0.reduce", "s1.pe1", "s1.pe2", "s1.reduce", "s2.pe1", "s2.pe2", "s2.reduce", "s3.pe1", "s3.pe2", "s3.reduce", "mt_proj"]
for nm, r, (kind, h) in list(zip(names, ref, hip))[::-1]:
    rg = r.grad
    if rg.dim() == 5: rg = rg.squeeze(2)
    rg = rg.permute(0, 2, 3, 1).reshape(-1)
    hg = h.grad.cpu().double().reshape(-1)
    print("  %-10s d(out) rel err %.2e   (max |ref| %.2e)" % (nm, (hg - rg).abs().max().item() / rg.abs().max().item(), rg.abs().max().item()))
worst = []
for n, p in net.named_parameters():
    g = leaf[n].grad
    if g is None or p.grad is None: continue
    worst.append(((p.grad.cpu().double() - g).abs().max().item() / (g.abs().max().item() + 1e-30), n))
worst.sort()
print("worst param grads:", [(f"{e:.1e}", n) for e, n in worst[-6:]])
print("median %.2e" % worst[len(worst) // 2][0])
