cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wino.py tests/test_gpu_salunet.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_sampling.py tests/test_gpu_lowp.py -x -q > $O/t_sal.log 2>&1
echo "rc=$?" >> $O/t_sal.log
tail -n 4 $O/t_sal.log
python bench.py --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg > $O/bench_fp32.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3w/bench_fp32.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'])
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('classes','note','kernel')})
for c in r.get('classes', d.get('classes', [])):
    print(c['class'], c['launches'], round(c['ms'],4), c['gflop'], c['frac'])
PY
