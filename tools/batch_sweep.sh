#!/bin/bash
# Batch sweep of the visual-only sampling bench on one GPU: one JSON line per (precision, batch) with the build id.
# usage: tools/batch_sweep.sh <out.jsonl>
OUT=$1; : > $OUT
BID=$(python3 $GRAFT_REPO_ROOT/tools/build_id.py)
for p in fp32 bf16; do for b in 1 2 4 8 16; do
python3 $GRAFT_REPO_ROOT/bench.py --batch $b --precision $p --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({'precision':'$p','batch':$b,'value':d['value'],'unit':d['unit'],'ms_per_step':d['ms_per_step'],'sampler_mode':d['config'].get('sampler_mode'),'build':'$BID'}))" >> $OUT
done; done
