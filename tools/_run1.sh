cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4
for b in 1 2 8; do echo "B=$b"; timeout 600 python tools/bench_wino.py $b 2>&1 | grep -v amdgpu.ids; done > gpurun_out/w4/bench_b.log 2>&1
cat gpurun_out/w4/bench_b.log
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/w4/t_all.log 2>&1
tail -n 15 gpurun_out/w4/t_all.log
for b in 4; do
timeout 600 python3 bench.py --batch $b --steps 100 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=$b', d['config']['sampler_mode'], d['value'], d['ms_per_step'])"
done
