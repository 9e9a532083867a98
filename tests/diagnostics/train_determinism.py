import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
cfg = CASES["tiny_av"][0]
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
x, feats, audio = orc.synth_inputs(cfg, 2, False, tag="train")
x0 = torch.sigmoid(orc.synth_tensor("train.x0", (2, 1, *cfg.img_size))).cuda()
t = torch.tensor([321, 321]).cuda()
net = build(cfg, sd); net.train(); net.dropout_p = 0.0
xd, fd = x.cuda(), [f.cuda() for f in feats]
runs = []
for r in range(4):
    net.zero_grad(set_to_none=True)
    out = net(xd, t, fd, None)
    l = ((out - x0) ** 2).sum(dim=(1, 2, 3)).mean(); l.backward(); torch.cuda.synchronize()
    runs.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    print("run", r, "loss", l.item())
bad = []
for n in runs[0]:
    d = max((runs[i][n] - runs[0][n]).abs().max().item() for i in range(1, 4))
    m = runs[0][n].abs().max().item()
    if d > 1e-4 * (m + 1e-6): bad.append((d / (m + 1e-12), n, d, m))
print("params with run-to-run differences > 1e-4 rel:", len(bad))
names = {b[1]: b[0] for b in bad}
for n in runs[0]:
    if any(k in n for k in ("mt_proj", "logits", "redu_chan_up", "norm_mts", "mid_stages.3", "mid_stages.2.blocks")):
        print(f"   {n:70s} {names.get(n, 0.0):.2e}")
