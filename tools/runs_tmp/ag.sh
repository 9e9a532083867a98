cd $GRAFT_REPO_ROOT
D=gpurun_out/r5ag; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_encoder_train.py -x -q -k "pool" > $D/tests.txt 2>&1
timeout 600 python3 tools/bench_pool.py > $D/pool.txt 2>&1
tail -5 $D/tests.txt; cat $D/pool.txt
