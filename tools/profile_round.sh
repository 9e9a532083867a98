#!/bin/bash
# Round profile of the default benchmark on the GPU box: rocprofv3 kernel-trace stats (CSV) and the two PMC passes for HBM
# traffic (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).  Writes under gpurun_out/$1.
# usage: tools/profile_round.sh <outdir-name> <precision> [extra bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; PREC=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-alt-precision --no-encoders --no-reference-graph --precision $PREC --steps 50 --warmup 5 --repeats 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$PREC -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/stats_$PREC.json 2> $OUT/stats_$PREC.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$PREC -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps 10 > /dev/null 2> $OUT/pmc_fetch_$PREC.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$PREC -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps 10 > /dev/null 2> $OUT/pmc_write_$PREC.err
find $OUT -name "*kernel_stats.csv" | head; find $OUT -name "*counter_collection.csv" | head
# keep only the summaries (the traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
# HBM traffic summary of the GEMM family (bench.py reads profiles/r02_igemm_hbm_traffic_<precision>.json)
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT/pmc_fetch_$PREC $OUT/pmc_write_$PREC $OUT/r02_igemm_hbm_traffic_$PREC.json $PREC "${BUILD_ID:-build n/a}"
cp $(find $OUT/stats_$PREC -name "*kernel_stats.csv" | head -1) $OUT/r02_${PREC}_kernel_stats.csv
