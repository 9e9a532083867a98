cd $GRAFT_REPO_ROOT
D=gpurun_out/r5b; mkdir -p $D
python3 tools/find_aten.py > $D/aten_fp32.txt 2>$D/aten_err.txt
python3 tools/find_aten.py --precision bf16 > $D/aten_bf16.txt 2>>$D/aten_err.txt
timeout 600 python -m pytest tests/test_gpu_upconv.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -5 > $D/tests.txt
cat $D/aten_fp32.txt; tail -3 $D/tests.txt
