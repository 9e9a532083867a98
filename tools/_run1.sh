cd $GRAFT_REPO_ROOT
export DIFFSAL_NO_REBUILD=1
for r in 1 2 3; do for v in A B; do
cp diff_sal_amd/lib$v.so diff_sal_amd/libdiffsal_hip.so
timeout 600 python3 bench.py --batch 4 --steps 100 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], [ (c['class'], c['ms']) for c in d['roofline'].get('classes', [])[:5]])
"
done; done
