cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
python bench.py --precision bf16 --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_bf16.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3m/bench_bf16.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d.get('end_to_end'))
PY
bash gpu_all.sh
