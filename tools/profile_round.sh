#!/bin/bash
# Round profile of the default benchmark on the GPU box, every artefact stamped with the source id (tools/build_id.py) of the
# code it was taken on: rocprofv3 kernel-trace stats (CSV), the per-launch table, and the two PMC passes for HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).  Writes under gpurun_out/<out>/ with the
# names profiles/ uses (<round>_...), so collecting is a copy.
# usage: [BATCH=64 STEPS=20 PSTEPS=5] tools/profile_round.sh <outdir-name> <round-tag e.g. r03> <precision> <mode vis|av> [pmc]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; TAG=$2; PREC=$3; MODE=$4; PMC=${5:-}
mkdir -p $OUT
BID=$(python3 $GRAFT_REPO_ROOT/tools/build_id.py)
SUF=${PREC}$([ -n "${BATCH:-}" ] && echo _b${BATCH})$([ "$MODE" = av ] && echo _av)
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-alt-precision --no-encoders --no-reference-graph --no-train-leg --precision $PREC --mode $MODE --warmup 5 --repeats 1$([ -n "${BATCH:-}" ] && echo " --batch ${BATCH}")"
STEPS=${STEPS:-50}; PSTEPS=${PSTEPS:-10}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$SUF -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps $STEPS --dump-launches $OUT/${TAG}_launches_$SUF.json > $OUT/${TAG}_bench_profiled_$SUF.json 2> $OUT/stats_$SUF.err
cp $(find $OUT/stats_$SUF -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${SUF}_kernel_stats.csv
echo "{\"file\": \"$(basename $OUT/${TAG}_${SUF}_kernel_stats.csv)\", \"build\": \"$BID\", \"command\": \"rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS --steps $STEPS\"}" >> $OUT/${TAG}_manifest.jsonl
if [ -n "$PMC" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$SUF -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps $PSTEPS > /dev/null 2> $OUT/pmc_fetch_$SUF.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$SUF -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps $PSTEPS > /dev/null 2> $OUT/pmc_write_$SUF.err
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT/pmc_fetch_$SUF $OUT/pmc_write_$SUF $OUT/${TAG}_hbm_traffic_$SUF.json $PREC "$BID" $OUT/${TAG}_launches_$SUF.json > $OUT/pmc_traffic_$SUF.log 2>&1
  # matrix-pipe busy fraction per kernel (its own pass; joined with the durations of the kernel-trace pass above)
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_busy_$SUF -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --steps $PSTEPS > /dev/null 2> $OUT/pmc_busy_$SUF.err
  { echo "<!-- build $BID: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python3 bench.py $ARGS --steps $PSTEPS, durations from the kernel-trace pass -->";
    python3 $GRAFT_REPO_ROOT/tools/pmc_mfma_busy.py $OUT/pmc_busy_$SUF $OUT/stats_$SUF 8; } > $OUT/${TAG}_pmc_mfma_busy_$SUF.md 2> $OUT/pmc_busy_tool_$SUF.err
  find $OUT -name "*counter_collection.csv" -size +5M -delete
fi
find $OUT -name "*kernel_trace.csv" -size +8M -delete
echo "profile_round $SUF done (build $BID)"
