cd $GRAFT_REPO_ROOT
D=gpurun_out/r5a; mkdir -p $D
for p in fp32 bf16; do for m in eager graph; do
python3 bench.py --steps 50 --warmup 10 --precision $p --sampler-mode $m --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph 2>>$D/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$p $m', d['value'], d['ms_per_step'], d['config'].get('sampler_mode'))" | tee -a $D/out.txt
done; done
