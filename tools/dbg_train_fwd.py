import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import salunet_oracle as orc
from tests._cases import CASES
from tests.test_gpu_salunet import build
cfg = CASES["tiny_av"][0]
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
t = torch.tensor([321, 321])
for av in (False, True):
    x, feats, audio = orc.synth_inputs(cfg, 2, av, tag="train")
    orc.BN_TRAIN = True
    with torch.no_grad():
        taps = {}
        pred = orc.salunet_forward(sd, cfg, x, t, feats, audio, taps=taps)
    orc.BN_TRAIN = False
    net = build(cfg, sd); net.train(); net.dropout_p = 0.0
    with torch.no_grad():
        out = net(x.cuda(), t.cuda(), [f.cuda() for f in feats], None if audio is None else audio.cuda())
    net.eval()
    with torch.no_grad():
        oe = net(x.cuda(), t.cuda(), [f.cuda() for f in feats], None if audio is None else audio.cuda())
        pe = orc.salunet_forward(sd, cfg, x, t, feats, audio)
    print("av", av, "train fwd max|d|", (out.cpu() - pred).abs().max().item(), " eval fwd max|d|", (oe.cpu() - pe).abs().max().item(),
          " multi_scale max", taps["multi_scale"].abs().max().item())
