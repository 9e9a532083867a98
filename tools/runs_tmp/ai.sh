cd $GRAFT_REPO_ROOT
D=gpurun_out/r5ai; mkdir -p $D
timeout 900 python3 tools/bench_wgrad_dma.py > $D/wgrad.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_train.py -x -q > $D/tests.txt 2>&1
for i in 1 2; do
    timeout 900 python3 bench.py --workload train --mode av --steps 10 --warmup 3 --no-cpu-baseline --no-solo-leg 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $D/train_ab.txt
done
cat $D/wgrad.txt; tail -3 $D/tests.txt; cat $D/train_ab.txt
