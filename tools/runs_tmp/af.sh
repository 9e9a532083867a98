cd $GRAFT_REPO_ROOT
D=gpurun_out/r5af; mkdir -p $D
python3 tools/bench_pair.py > $D/pair.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_lowp.py -x -q > $D/tests.txt 2>&1
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2 3; do
    $B --precision bf16 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bf16', d['value'], d['ms_per_step'])" >> $D/ab.txt
done
$B --precision fp16 --batch 64 --mode av --steps 20 --warmup 5 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fp16-64', d['value'], d['ms_per_step'])" >> $D/ab.txt
cat $D/pair.txt; tail -2 $D/tests.txt; cat $D/ab.txt
