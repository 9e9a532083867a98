import os, sys, torch, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops
shapes = [((4,112,192,768),96,3,1), ((36,28,48,384),192,3,1), ((36,14,24,768),384,3,1), ((36,56,96,192),96,3,1), ((36,56,96,96),96,3,1),
          ((1,1,193536,96),96,1,0), ((1,1,193536,96),192,1,0), ((1,1,48384,192),192,1,0), ((4,9,5376,96),768,(5,1),(0,0))]
dev = "cuda"
for xs, co, k, pad in shapes:
    x = torch.randn(xs, device=dev)
    kh, kw = (k, k) if isinstance(k, int) else k
    pd = (pad, pad) if isinstance(pad, int) else pad
    st = (1, 1)
    if kh == 5:
        st = (5, 1); dy = torch.randn((xs[0], 1, xs[2], co), device=dev)
    else:
        dy = torch.randn((xs[0], xs[1], xs[2], co), device=dev)
    fl = 2.0 * dy.numel() * kh * kw * xs[3]
    for _ in range(3): ops.conv_wgrad(x, dy, kh=kh, kw=kw, pad=pd, stride=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv_wgrad(x, dy, kh=kh, kw=kw, pad=pd, stride=st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("cfg=%s wgs=%s  x%s co=%d k%dx%d : %8.1f us %6.1f TF" % (os.environ.get("DIFFSAL_WGRAD_CFG", "auto"), os.environ.get("DIFFSAL_WGRAD_WGS", "1024"), xs, co, kh, kw, us, fl / us / 1e6))
