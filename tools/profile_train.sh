#!/bin/bash
# Training-step profile (BASELINE configs[3], full model, AV, B=4): bench line + per-launch table + rocprofv3 kernel stats, stamped
# with the source id.  usage: tools/profile_train.sh <outdir-name> <round-tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; TAG=$2
mkdir -p $OUT
BID=$(python3 $GRAFT_REPO_ROOT/tools/build_id.py)
cd $GRAFT_REPO_ROOT
python3 bench.py --workload train --mode av --steps 10 --warmup 3 --repeats 3 --dump-launches $OUT/${TAG}_launches_train.json > $OUT/${TAG}_train_full.json 2> $OUT/train.err
python3 bench.py --workload train --mode av --train-scope decoder --steps 20 --warmup 3 --repeats 3 > $OUT/${TAG}_train_decoder.json 2>> $OUT/train.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --mode av --steps 10 --warmup 3 --repeats 1 > /dev/null 2> $OUT/stats_train.err
cp $(find $OUT/stats_train -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_train_kernel_stats.csv
echo "{\"file\": \"$(basename $OUT/${TAG}_train_kernel_stats.csv)\", \"build\": \"$BID\", \"command\": \"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --mode av --steps 10 --warmup 3 --repeats 1 (13 steps)\"}" >> $OUT/${TAG}_manifest.jsonl
find $OUT -name "*kernel_trace.csv" -size +8M -delete
echo "profile_train done (build $BID)"
