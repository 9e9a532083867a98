# scratch: the command of the last ad-hoc GPU call (see tools/_run_all.sh, tools/gpu_prof.sh for the full suite / evidence runs)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_upconv.py tests/test_gpu_wino.py tests/test_gpu_gemm_dma.py -x -q 2>&1 | tail -3
