cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_salunet.py tests/test_gpu_fullsize.py tests/test_legacy_unet.py -q -m gpu 2>&1 | tail -n 4
timeout 900 python3 bench.py --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])
for c in d['roofline']['classes']:
    if c['class'] in ('K3',): print(c['class'], c['launches'], c['ms'], c['gbs'])
"
