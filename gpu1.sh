cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout 2400 python -m pytest tests/test_gpu_encoders.py tests/test_gpu_encoder_train.py tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -x -q > gpurun_out/r3m/t.log 2>&1
echo "rc=$?" >> gpurun_out/r3m/t.log
timeout 900 python bench.py --workload train --mode av --steps 10 --warmup 3 --repeats 3 --dump-launches gpurun_out/r3m/launches_train.json > gpurun_out/r3m/train.json 2> gpurun_out/r3m/train.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precision > gpurun_out/r3m/bench_fp32.json 2> gpurun_out/r3m/bench.err
tail -n 4 gpurun_out/r3m/t.log
