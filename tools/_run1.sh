cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 600 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "reference_loss_dictionary" > gpurun_out/r4a/t6.log 2>&1
tail -3 gpurun_out/r4a/t6.log
timeout 600 python3 tools/bench_gemm_dma.py b1 --cfgs 1 2>&1 | tail -15
