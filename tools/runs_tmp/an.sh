cd $GRAFT_REPO_ROOT
D=gpurun_out/r5an; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_encoder_train.py tests/test_gpu_train_step.py -x -q > $D/tests.txt 2>&1
for i in 1 2; do
    timeout 900 python3 bench.py --workload train --mode av --steps 10 --warmup 3 --no-cpu-baseline --no-solo-leg 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $D/train_ab.txt
done
tail -4 $D/tests.txt; cat $D/train_ab.txt
