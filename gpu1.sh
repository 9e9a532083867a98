cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout 2400 python -m pytest tests/test_gpu_encoders.py tests/test_gpu_encoder_train.py -x -q > gpurun_out/r3n/t.log 2>&1
echo "rc=$?" >> gpurun_out/r3n/t.log
timeout 900 python bench.py --workload train --mode av --steps 10 --warmup 3 --repeats 3 --dump-launches gpurun_out/r3n/launches_train.json > gpurun_out/r3n/train.json 2> gpurun_out/r3n/train.err
tail -n 4 gpurun_out/r3n/t.log
