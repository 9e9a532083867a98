cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3r
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_salunet.py tests/test_gpu_lowp.py -x -q -k "tapsum or golden or tap" > gpurun_out/r3r/t.log 2>&1
echo "rc=$?" >> gpurun_out/r3r/t.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --dump-launches gpurun_out/r3r/launches_fp32.json > gpurun_out/r3r/bench_fp32.json 2> gpurun_out/r3r/bench.err
tail -n 3 gpurun_out/r3r/t.log
