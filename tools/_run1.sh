cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4
timeout 900 python -m pytest tests/test_gpu_upconv.py -x -q -s 2>&1 | grep -v amdgpu | tail -25
timeout 3000 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_fullsize.py tests/test_gpu_wino.py tests/test_gpu_sampling.py -x -q > gpurun_out/w4/t_part.log 2>&1
tail -n 5 gpurun_out/w4/t_part.log
for i in 1 2; do
timeout 600 python3 bench.py --batch 4 --steps 100 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph --dump-launches gpurun_out/w4/launches.json 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=4', d['config']['sampler_mode'], d['value'], d['ms_per_step'], [ (c['class'], c['launches'], c['ms']) for c in d['roofline'].get('classes', [])[:9]])
"
done
