cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 1200 python -m pytest tests/test_gpu_block_front.py tests/test_gpu_ops.py tests/test_gpu_lowp.py -x -q -k "block_front or mlp_block or block16 or kv_prep" > gpurun_out/r3e/t.log 2>&1
echo "rc=$?" >> gpurun_out/r3e/t.log
timeout 300 python tools/bench_block_front.py > gpurun_out/r3e/bf.log 2>&1
DIFFSAL_EXTRA_HIPCC_FLAGS=-DDIFFSAL_DEV_STAMPS DIFFSAL_BUILD_JOBS=16 python -m diff_sal_amd.build --force > gpurun_out/r3e/build.log 2>&1
python tools/probe_front_stamps.py > gpurun_out/r3e/stamps.log 2>&1
tail -n 3 gpurun_out/r3e/t.log; cat gpurun_out/r3e/bf.log gpurun_out/r3e/stamps.log
