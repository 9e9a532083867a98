cd $GRAFT_REPO_ROOT
D=gpurun_out/r5g; mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $D/tests_all.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph --precision bf16"
for i in 1 2; do
DIFFSAL_NO_GN_SLAB=1 $B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bf16 two-launch GN', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bf16 slab GN      ', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
tail -5 $D/tests_all.txt
