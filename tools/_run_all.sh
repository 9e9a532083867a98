cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r4b/t_all.log 2>&1
echo "rc=$?" >> gpurun_out/r4b/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4b/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r4b/smoke.log
for b in 1 2 4; do
timeout 600 python3 bench.py --batch $b --steps 100 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=$b', d['config']['sampler_mode'], d['value'], d['ms_per_step'])" >> gpurun_out/r4b/batch.log
done
tail -n 5 gpurun_out/r4b/t_all.log; tail -n 3 gpurun_out/r4b/smoke.log; cat gpurun_out/r4b/batch.log
