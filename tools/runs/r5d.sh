cd $GRAFT_REPO_ROOT
D=gpurun_out/r5d; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q -k "resblock or gn_affine" 2>&1 | tail -5 > $D/tests1.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2; do
$B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fused  ', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B --set fuse_resblock=0 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('unfused', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
$B --dump-launches $D/launches_fp32.json > $D/bench_fp32.json 2>>$D/err.txt
tail -3 $D/tests1.txt
