#!/bin/bash
# Same-box A/B of the sampling bench: alternates the variants ROUNDS times inside ONE gpurun call (box-to-box spread is +-0.5 %, larger
# than most kernel-level effects) and prints value / ms_per_step per run plus the per-variant median.
# usage: tools/ab_bench.sh [-r ROUNDS] [-o outdir-under-gpurun_out] [-c "common bench args"] "label=ENV1=v ENV2=v -- extra bench args" ...
#   e.g. tools/ab_bench.sh -c "--precision bf16" "plain=" "wide tiles=DIFFSAL_BATCH_TILE=1 --" "unfused=-- --set fuse_resblock=0"
R=3; OUT=ab; COMMON=""
while getopts "r:o:c:" o; do case $o in r) R=$OPTARG;; o) OUT=$OPTARG;; c) COMMON=$OPTARG;; esac; done
shift $((OPTIND - 1))
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
D=gpurun_out/$OUT; mkdir -p $D; : > $D/ab.txt
BASE="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph $COMMON"
for i in $(seq $R); do
  for v in "$@"; do
    label=${v%%=*}; spec=${v#*=}
    envs=${spec%%--*}; args=""; [[ "$spec" == *"--"* ]] && args=${spec#*--}
    line=$(env $envs $BASE $args 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
    echo "$label|$line" | tee -a $D/ab.txt
  done
done
python3 - $D/ab.txt <<'PY'
import sys, statistics
from collections import defaultdict
d = defaultdict(list)
for ln in open(sys.argv[1]):
    lab, rest = ln.rstrip("\n").split("|", 1)
    if rest.strip():
        d[lab].append(float(rest.split()[1]))
for lab, v in d.items():
    print(f"{lab:30s} median {statistics.median(v):.4f} ms  ({len(v)} runs: {' '.join(f'{x:.4f}' for x in v)})")
PY
