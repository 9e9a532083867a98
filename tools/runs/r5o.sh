cd $GRAFT_REPO_ROOT
D=gpurun_out/r5o; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_lowp.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -40 > $D/tests1.txt
cat $D/tests1.txt
