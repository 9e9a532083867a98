cd $GRAFT_REPO_ROOT
D=gpurun_out/r5r; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_salunet.py tests/test_gpu_lowp.py -x -q -k "audio or av or golden" 2>&1 | tail -5 > $D/tests1.txt
B="python3 bench.py --steps 50 --warmup 10 --mode av --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
$B --dump-launches $D/launches_av.json 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('av fp32', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B --precision bf16 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('av bf16', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
tail -3 $D/tests1.txt
python3 -c "
import json
for l in json.load(open('$D/launches_av.json'))['launches']:
    if l['class'].startswith('K7'): print(l['class'], l['us'])"
