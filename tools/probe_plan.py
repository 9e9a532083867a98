import os, sys, torch
sys.path.insert(0, os.getcwd())
from diff_sal_amd import ops
from tools.tune_igemm16 import timed
for M, K, N in [(48384, 192, 864), (12096, 384, 1728)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
    for env in [{}, {"DIFFSAL_IGEMM_CFG": "5"}, {"DIFFSAL_IGEMM_CFG": "2"}, {"DIFFSAL_NO_XCD_ORDER": "1"}, {"DIFFSAL_IGEMM_CFG": "5", "DIFFSAL_NO_XCD_ORDER": "1"}]:
        for k in ("DIFFSAL_IGEMM_CFG", "DIFFSAL_NO_XCD_ORDER"): os.environ.pop(k, None)
        os.environ.update(env)
        ops.linear(x, w, None)
        print(M, K, N, env, round(timed(lambda: ops.linear(x, w, None)), 1))
