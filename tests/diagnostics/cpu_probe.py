import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import salunet_oracle as orc
cfg = orc.SalUNetConfig()
sd = orc.synth_state_dict(orc.state_dict_template(cfg))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for B in (1, 4):
    x, feats, _ = orc.synth_inputs(cfg, B, False)
    t = torch.full((B,), 500.0)
    for nt in (16, 32, 64, 128):
        torch.set_num_threads(nt)
        with torch.no_grad():
            orc.salunet_forward(sd, cfg, x[:1], t[:1], [f[:1] for f in feats])
            t0 = time.perf_counter(); orc.salunet_forward(sd, cfg, x, t, feats); dt = time.perf_counter() - t0
        print(f"B={B} threads={nt}: {dt:.2f} s/eval -> {B/dt:.3f} steps/s", flush=True)
