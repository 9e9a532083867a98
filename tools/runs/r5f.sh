cd $GRAFT_REPO_ROOT
D=gpurun_out/r5f; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_gemm_dma.py tests/test_gpu_upconv.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -8 > $D/tests1.txt
timeout 900 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5 > $D/tests2.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2; do
DIFFSAL_NO_XCD_ORDER=1 $B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('plain order', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('xcd order  ', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
tail -3 $D/tests1.txt; tail -3 $D/tests2.txt
