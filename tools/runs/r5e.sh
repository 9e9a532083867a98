cd $GRAFT_REPO_ROOT
D=gpurun_out/r5e; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tapsum" 2>&1 | tail -8 > $D/tests1.txt
timeout 900 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 > $D/tests2.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2; do
DIFFSAL_NO_TAPSUM_HEAD4=1 $B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('general', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('head4  ', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
$B --dump-launches $D/launches_fp32.json > $D/bench_fp32.json 2>>$D/err.txt
tail -3 $D/tests1.txt; tail -3 $D/tests2.txt
python3 -c "
import json
for l in json.load(open('$D/launches_fp32.json'))['launches']:
    if l['class'] in ('K14-tap',): print(l['class'], l['us'])"
