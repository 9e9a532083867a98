cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
for p in bf16 fp16; do
timeout 900 python3 bench.py --precision $p --no-cpu-baseline --no-encoders --no-train-leg > gpurun_out/r4b/bench_$p.json 2> gpurun_out/r4b/bench_$p.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r4b/bench_$p.json').read().strip().splitlines()[-1]); print('$p', d['value'], d['ms_per_step'])
for k in d['roofline']['kernels'][:6]: print(k)
"
done
DIFFSAL_GEMM_DMA16=0 timeout 900 python3 bench.py --precision bf16 --no-cpu-baseline --no-encoders --no-train-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bf16 dma16 off', d['value'], d['ms_per_step'])"
timeout 1200 python -m pytest tests/test_gpu_lowp.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu > gpurun_out/r4b/t8.log 2>&1
tail -n 4 gpurun_out/r4b/t8.log
