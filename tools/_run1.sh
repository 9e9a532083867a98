cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4
timeout 600 python3 bench.py --batch 4 --precision bf16 --steps 50 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph --dump-launches gpurun_out/w4/l16.json > /dev/null 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/w4/l16.json'))
for e in d['launches']:
    if e['class'].startswith('K12'): print(e['class'], e['us'], e['op'][:70], '|', e['kernel'][:60])
PY
