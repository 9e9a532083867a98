cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_gpu_block_front.py -x -q > gpurun_out/r3h/t.log 2>&1
timeout 300 python tools/bench_block_front.py > gpurun_out/r3h/bf_ahead.log 2>&1
tail -n 2 gpurun_out/r3h/t.log; cat gpurun_out/r3h/bf_ahead.log
