import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
from diff_sal_amd import autograd_ops as ag, ops
cfg = CASES["tiny_av"][0]
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
t = torch.tensor([321, 321])
_orig_emb = orc.timestep_embedding
orc.timestep_embedding = lambda tt, d: _orig_emb(tt, d).double()
rec = {}
orig_rs = ag.resize_sum
def spy(xs, H, W):
    y = orig_rs(xs, H, W); y.retain_grad(); rec["acc"] = y; return y
ag.resize_sum = spy
for av in (False, True):
    x, feats, audio = orc.synth_inputs(cfg, 2, av, tag="train")
    x0 = torch.sigmoid(orc.synth_tensor("train.x0", (2, 1, *cfg.img_size)))
    leaf = {k: (v.double() if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    orc.BN_TRAIN = True
    taps = {}
    pred = orc.salunet_forward(leaf, cfg, x.double(), t, [f.double() for f in feats], None if audio is None else audio.double(), taps=taps)
    orc.BN_TRAIN = False
    taps["multi_scale"].retain_grad()
    ((pred - x0.double()) ** 2).sum(dim=(1, 2, 3)).mean().backward()
    net = build(cfg, sd); net.train(); net.dropout_p = 0.0
    out = net(x.cuda(), t.cuda(), [f.cuda() for f in feats], None if audio is None else audio.cuda())
    ((out - x0.cuda()) ** 2).sum(dim=(1, 2, 3)).mean().backward()
    a_ref = taps["multi_scale"].detach().permute(0, 2, 3, 1)
    g_ref = taps["multi_scale"].grad.permute(0, 2, 3, 1)
    a, g = rec["acc"].detach().cpu().double(), rec["acc"].grad.cpu().double()
    print("av", av, "acc fwd err %.2e  d(acc) err %.2e" % ((a - a_ref).abs().max().item() / a_ref.abs().max().item(), (g - g_ref).abs().max().item() / g_ref.abs().max().item()))
    w = dict(net.named_parameters())["invpt_decoder.mt_proj.0.weight"]
    print("     mt_proj.0.weight grad err %.2e" % ((w.grad.cpu().double() - leaf["invpt_decoder.mt_proj.0.weight"].grad).abs().max().item() / leaf["invpt_decoder.mt_proj.0.weight"].grad.abs().max().item()))
