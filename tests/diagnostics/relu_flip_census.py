"""Count ReLU sign disagreements between the HIP train forward and the fp64 oracle (debug aid)."""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
from diff_sal_amd import autograd_ops as ag
cfg = orc.SalUNetConfig() if "full" in sys.argv else CASES["tiny_av"][0]
B = 1 if "full" in sys.argv else 2
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
tag, t0 = sys.argv[1], int(sys.argv[2])
av = "av" in sys.argv
x, feats, audio = orc.synth_inputs(cfg, B, av, tag=tag)
t = torch.tensor([t0] * B)
if "qs" in sys.argv:
    from diff_sal_amd.diffusion_utils import get_beta_schedule, to_torch
    a_hat = (1.0 - to_torch(get_beta_schedule("cosine", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000))).cumprod(dim=0)
    sal = torch.sigmoid(orc.synth_tensor(tag + ".sal", (2, 1, *cfg.img_size)))
    x0 = sal + 0.01 * orc.synth_tensor(tag + ".dq", tuple(sal.shape))
    x = a_hat[t0].sqrt() * x0 + (1 - a_hat[t0]).sqrt() * x
_orig_emb = orc.timestep_embedding
orc.timestep_embedding = lambda tt, d: _orig_emb(tt, d).double()
ref_relu = []
orig_relu = F.relu
def spy_relu(v, *a, **k):
    y = orig_relu(v, *a, **k); ref_relu.append(v.detach()); return y
orc.F.relu = spy_relu
orc.BN_TRAIN = True
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
with torch.no_grad():
    orc.salunet_forward(sd64, cfg, x.double(), t, [f.double() for f in feats], None if audio is None else audio.double())
orc.BN_TRAIN = False
orc.F.relu = orig_relu
hip = []
o_bn, o_conv = ag.batchnorm_relu_train, ag.conv
def spy_bn(x_, bn, relu=True):
    y = o_bn(x_, bn, relu); hip.append(("bn", y.detach())); return y
def spy_conv(x_, w, **k):
    y = o_conv(x_, w, **k)
    if k.get("act", 0) == 1: hip.append(("conv", y.detach()))
    return y
ag.batchnorm_relu_train, ag.conv = spy_bn, spy_conv
net = build(cfg, sd); net.train(); net.dropout_p = 0.0
with torch.enable_grad():
    net(x.cuda(), t.cuda(), [f.cuda() for f in feats], None if audio is None else audio.cuda())
print(len(ref_relu), len(hip))
for r, (kind, h) in zip(ref_relu, hip):
    if r.dim() == 5: r = r.squeeze(2)
    r = r.permute(0, 2, 3, 1).reshape(-1)
    h = h.cpu().reshape(-1)
    flips = ((r > 0) != (h > 0)).sum().item()
    near = (r.abs() < 1e-5).sum().item()
    print("  %-5s n=%8d  mask flips %d  |pre|<1e-5: %d  min|pre| %.2e" % (kind, r.numel(), flips, near, r.abs().min().item()))
