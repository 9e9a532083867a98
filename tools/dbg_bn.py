import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
from diff_sal_amd import autograd_ops as ag
cfg = CASES["tiny_av"][0]
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
t = torch.tensor([321, 321]).cuda()
rec = []
orig = ag.batchnorm_relu_train
def spy(x, bn, relu=True):
    x = x.detach().requires_grad_(True) if not x.requires_grad else x
    x.retain_grad()
    y = orig(x, bn, relu)
    y.retain_grad()
    rec.append((x, y, bn))
    return y
ag.batchnorm_relu_train = spy
for av in (False, True):
    rec.clear()
    x, feats, audio = orc.synth_inputs(cfg, 2, av, tag="train")
    x0 = torch.sigmoid(orc.synth_tensor("train.x0", (2, 1, *cfg.img_size))).cuda()
    net = build(cfg, sd); net.train(); net.dropout_p = 0.0
    out = net(x.cuda(), t, [f.cuda() for f in feats], None if audio is None else audio.cuda())
    ((out - x0) ** 2).sum(dim=(1, 2, 3)).mean().backward()
    for i, (xi, yi, bn) in enumerate(rec):
        # reference BN+ReLU backward in fp64 from the SAME x and dy
        xr = xi.detach().cpu().double().requires_grad_(True)
        C = xr.shape[-1]
        g, b = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
        x2 = xr.reshape(-1, C)
        yr = F.relu(F.batch_norm(x2.t().unsqueeze(0), None, None, g, b, True, 0.0, bn.eps).squeeze(0).t())
        yr.backward(yi.grad.detach().cpu().double().reshape(-1, C))
        e = (xi.grad.cpu().double().reshape(-1, C) - xr.grad.reshape(-1, C)).abs().max().item() / xr.grad.abs().max().item()
        ef = (yi.detach().cpu().double().reshape(-1, C) - yr.detach()).abs().max().item() / yr.abs().max().item()
        print("av", av, "BN#", i, tuple(xi.shape), "fwd err %.2e  dx err %.2e  |dy|max %.3g |dx|max %.3g" % (ef, e, yi.grad.abs().max().item(), xr.grad.abs().max().item()))
