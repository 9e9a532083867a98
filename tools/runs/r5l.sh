cd $GRAFT_REPO_ROOT
D=gpurun_out/r5l; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_upconv.py tests/test_gpu_wino.py -x -q 2>&1 | tail -12 > $D/tests1.txt
timeout 900 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 > $D/tests2.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
run() { "$@" 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "two-step  $(run $B --set fuse_up_pe2=0)" | tee -a $D/ab.txt
echo "fused     $(run $B)" | tee -a $D/ab.txt
done
$B --dump-launches $D/launches_fp32.json > $D/bench_fp32.json 2>>$D/err.txt
tail -5 $D/tests1.txt; tail -4 $D/tests2.txt
