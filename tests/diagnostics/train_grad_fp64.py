"""Which is closer to the fp64 gradient: the fp32 CPU oracle or the HIP path?  (AV mode, tiny config)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
cfg = CASES["tiny_av"][0]
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
x, feats, audio = orc.synth_inputs(cfg, 2, True, tag="train")
x0 = torch.sigmoid(orc.synth_tensor("train.x0", (2, 1, *cfg.img_size)))
t = torch.tensor([321, 321])
_orig_emb = orc.timestep_embedding
def ref(dtype):
    orc.timestep_embedding = lambda tt, d: _orig_emb(tt, d).to(dtype)
    leaf = {k: (v.to(dtype) if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    orc.BN_TRAIN = True
    pred = orc.salunet_forward(leaf, cfg, x.to(dtype), t, [f.to(dtype) for f in feats], audio.to(dtype))
    orc.BN_TRAIN = False
    ((pred - x0.to(dtype)) ** 2).sum(dim=(1, 2, 3)).mean().backward()
    return {k: v.grad for k, v in leaf.items() if v.grad is not None}
g64, g32 = ref(torch.float64), ref(torch.float32)
net = build(cfg, sd); net.train(); net.dropout_p = 0.0
out = net(x.cuda(), t.cuda(), [f.cuda() for f in feats], audio.cuda())
((out - x0.cuda()) ** 2).sum(dim=(1, 2, 3)).mean().backward()
import statistics
e32, ehip, rows = [], [], []
for n, p in net.named_parameters():
    if n not in g64 or p.grad is None: continue
    m = g64[n].abs().max().item() + 1e-12
    e32.append((g32[n].double() - g64[n]).abs().max().item() / m)
    ehip.append((p.grad.cpu().double() - g64[n]).abs().max().item() / m)
    rows.append((ehip[-1], n, m))
sel = ["logits.linear_pred.weight", "invpt_decoder.mt_proj.1.weight", "invpt_decoder.mt_proj.0.weight",
       "invpt_decoder.redu_chan_up.3.proj.0.weight", "invpt_decoder.redu_chan_up.0.proj.0.weight", "invpt_decoder.norm_mts.3.weight",
       "invpt_decoder.mid_stages.3.blocks.0.mlp.fc2.weight", "invpt_decoder.mid_stages.3.blocks.0.mlp.fc1.weight",
       "invpt_decoder.mid_stages.3.blocks.0.norm2.weight", "invpt_decoder.mid_stages.3.blocks.0.attn.proj.weight",
       "invpt_decoder.mid_stages.3.blocks.0.attn.proj_v.weight", "invpt_decoder.mid_stages.3.blocks.0.attn.proj_q.weight",
       "invpt_decoder.mid_stages.3.blocks.0.norm.weight", "invpt_decoder.mid_stages.3.patch_embed.0.proj.5.weight"]
d = {r[1]: r[0] for r in rows}
print("backward order:"); [print("   %.2e %s" % (d.get(n, -1), n)) for n in sel]
rows.sort()
print("lowest HIP errors:"); [print("   %.2e %s %.3g" % r) for r in rows[:12]]
print("highest HIP errors:"); [print("   %.2e %s %.3g" % r) for r in rows[-25:]]
print("median rel err vs fp64: cpu-fp32 oracle %.2e   HIP %.2e" % (statistics.median(e32), statistics.median(ehip)))
print("p90: cpu-fp32 %.2e   HIP %.2e" % (sorted(e32)[int(.9*len(e32))], sorted(ehip)[int(.9*len(ehip))]))
