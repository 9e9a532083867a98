cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_gemm_dma.py -x -q -m gpu > gpurun_out/r4a/t2.log 2>&1
tail -5 gpurun_out/r4a/t2.log
( time timeout 1500 python3 bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err ) 2>&1 | grep real
python3 -c "
import json
d=json.loads(open('gpurun_out/r4a/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','frac_reference_graph','dominant_kernel')})
print(d['cpu_baseline'])
for k in r['kernels']: print(k)
print(d.get('train',{}).get('value'), d.get('train',{}).get('ms_per_step'))
"
