cd $GRAFT_REPO_ROOT
D=gpurun_out/r5ab; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_encoder_train.py tests/test_gpu_encoders.py -x -q -k "attention" > $D/tests.txt 2>&1
timeout 600 python3 tools/bench_attn_bwd.py > $D/attn_bwd.txt 2>&1
for i in 1 2; do
    timeout 900 python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline --no-solo-leg 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $D/train_ab.txt
done
cat $D/tests.txt | tail -3; cat $D/attn_bwd.txt; cat $D/train_ab.txt
