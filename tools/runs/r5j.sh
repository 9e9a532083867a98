cd $GRAFT_REPO_ROOT
D=gpurun_out/r5j; mkdir -p $D
B="python3 bench.py --steps 20 --warmup 5 --precision fp16 --batch 64 --mode av --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2; do
DIFFSAL_NO_GN_SLAB=1 $B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fp16 B=64 two-launch GN', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
$B 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('fp16 B=64 slab GN      ', d['value'], d['ms_per_step'])" | tee -a $D/ab.txt
done
