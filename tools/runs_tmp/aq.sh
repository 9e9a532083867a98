cd $GRAFT_REPO_ROOT
D=gpurun_out/r5aq; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_train_ops.py -x -q -s -k "winograd" > $D/tests0.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_gpu_fullsize.py -x -q > $D/tests.txt 2>&1
for i in 1 2; do
  for w in 1 0; do
    DIFFSAL_NO_WINO4_TRAIN=$w timeout 900 python3 bench.py --workload train --mode av --steps 10 --warmup 3 --no-cpu-baseline --no-solo-leg 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('no_wino=$w', d['value'], d['ms_per_step'])" >> $D/train_ab.txt
  done
done
grep -E "winograd vs|passed|failed" $D/tests0.txt | tail -20; tail -4 $D/tests.txt; cat $D/train_ab.txt
