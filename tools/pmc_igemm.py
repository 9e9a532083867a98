"""One big convolution (stage-3 UpEmbed conv1 shape) a few times, for rocprofv3 --pmc passes.  DIFFSAL_PRECISION selects the mode."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sal_amd import ops
ops.set_gemm_precision(os.environ.get("DIFFSAL_PRECISION", "fp32"))
x = torch.relu(torch.randn(36, 56, 96, 192, device="cuda")); w = torch.randn(96, 9 * 192, device="cuda") * 0.05
for _ in range(4): ops.conv_igemm(x, w, kh=3, kw=3, pad=(2, 2), dil=(2, 2))
torch.cuda.synchronize()
