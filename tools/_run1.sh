cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -2
for m in eager graph eager graph; do
timeout 600 python3 bench.py --batch 4 --steps 100 --sampler-mode $m --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=4', d['config']['sampler_mode'], d['value'], d['ms_per_step'])"
done
