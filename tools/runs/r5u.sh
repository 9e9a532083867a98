cd $GRAFT_REPO_ROOT
D=gpurun_out/r5u; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "tapsum" 2>&1 | tail -3 > $D/tests1.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
run() { "$@" 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "two images per wave (2 sets)  $(run $B)" | tee -a $D/ab.txt
echo "one image per wave (3 sets)   $(DIFFSAL_TAPSUM_ROWS_FORM=2 run $B)" | tee -a $D/ab.txt
done
DIFFSAL_TAPSUM_ROWS_FORM=2 $B --dump-launches $D/l2.json > /dev/null 2>>$D/err.txt
python3 -c "
import json
for l in json.load(open('$D/l2.json'))['launches']:
    if l['class'] in ('K14-tap',): print(l['class'], l['us'])"
tail -2 $D/tests1.txt
