cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -m gpu -x -q -k "fullsize or eval_after_training or tile_shape or persistent_linear or halo_conv or linear_pair" > gpurun_out/r3a/tests_new.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r3a/tests_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
echo "bench rc=$?" >> gpurun_out/r3a/bench.err
tail -5 gpurun_out/r3a/tests_new.log
