cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_gemm_dma.py tests/test_gpu_salunet.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4a/t4.log 2>&1
tail -5 gpurun_out/r4a/t4.log
timeout 900 python3 bench.py --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg > gpurun_out/r4a/bench_fp32c.json 2> gpurun_out/r4a/bench_fp32c.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r4a/bench_fp32c.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for k in d['roofline']['kernels']: print(k)
for c in d['roofline']['classes']: print(c['class'], c['launches'], c['ms'])
"
