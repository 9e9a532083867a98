cd $GRAFT_REPO_ROOT
D=gpurun_out/r5ac; mkdir -p $D
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for i in 1 2; do
for m in eager graph; do
  for p in bf16 fp32; do
    $B --precision $p --sampler-mode $m 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$p $m', d['value'], d['ms_per_step'])" >> $D/ab.txt
  done
done
done
cat $D/ab.txt
