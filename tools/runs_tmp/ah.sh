cd $GRAFT_REPO_ROOT
D=gpurun_out/r5ah; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_encoder_train.py -x -q > $D/tests.txt 2>&1
timeout 600 python3 tools/bench_pool.py > $D/pool.txt 2>&1
for i in 1 2; do
  for v in 1 0; do
    DIFFSAL_NO_POOL_RUNS=$v timeout 900 python3 bench.py --workload train --mode av --steps 10 --warmup 3 --no-cpu-baseline --no-solo-leg 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('no_runs=$v', d['value'], d['ms_per_step'])" >> $D/train_ab.txt
  done
done
tail -3 $D/tests.txt; cat $D/pool.txt; cat $D/train_ab.txt
