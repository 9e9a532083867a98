cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_gemm_dma.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r4a/t3.log 2>&1
tail -5 gpurun_out/r4a/t3.log
timeout 900 python3 bench.py --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg > gpurun_out/r4a/bench_fp32b.json 2> gpurun_out/r4a/bench_fp32b.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r4a/bench_fp32b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for k in d['roofline']['kernels']: print(k)
"
