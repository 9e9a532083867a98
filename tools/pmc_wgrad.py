"""One token-GEMM weight gradient (K = Cout = 96, M = 193536) a few times, for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops
M, K, Co = 193536, int(os.environ.get("PK", 96)), int(os.environ.get("PCO", 96))
x = torch.randn((1, 1, M, K), device="cuda"); dy = torch.randn((1, 1, M, Co), device="cuda")
for _ in range(4): ops.conv_wgrad(x, dy)
torch.cuda.synchronize()
