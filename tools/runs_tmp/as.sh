cd $GRAFT_REPO_ROOT
D=gpurun_out/r5as; mkdir -p $D
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
for b in 1 2 4; do
  for i in 1 2; do
    $B --batch $b 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('B=$b', d['value'], d['ms_per_step'])" >> $D/ab.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_sampling.py -x -q 2>&1 | tail -2 >> $D/ab.txt
cat $D/ab.txt
