cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
python bench.py --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph > $O/bench_fp32_t.json 2> $O/bench.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r3w/bench_fp32_t.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_all_regions'))
for c in d['roofline'].get('classes', [])[:5]:
    print(c['class'], c['launches'], round(c['ms'],4), c['gflop'], c['frac'])
PY
timeout 1200 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -3
