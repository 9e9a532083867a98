cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 2400 python -m pytest tests/test_gpu_block_front.py tests/test_gpu_ops.py tests/test_gpu_salunet.py -x -q > gpurun_out/r3j/t.log 2>&1
echo "rc=$?" >> gpurun_out/r3j/t.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --dump-launches gpurun_out/r3j/launches_fp32.json > gpurun_out/r3j/bench_fp32.json 2> gpurun_out/r3j/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-encoders --precision bf16 --dump-launches gpurun_out/r3j/launches_bf16.json > gpurun_out/r3j/bench_bf16.json 2>> gpurun_out/r3j/bench.err
tail -n 4 gpurun_out/r3j/t.log
