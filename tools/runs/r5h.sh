cd $GRAFT_REPO_ROOT
D=gpurun_out/r5h; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tapsum" 2>&1 | tail -8 > $D/tests1.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2; do
DIFFSAL_NO_TAPSUM_ROWS=1 $B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('general', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rows   ', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
$B --dump-launches $D/launches_fp32.json > $D/bench_fp32.json 2>>$D/err.txt
tail -3 $D/tests1.txt
python3 -c "
import json
for l in json.load(open('$D/launches_fp32.json'))['launches']:
    if l['class'] in ('K14-tap',): print(l['class'], l['us'])"
