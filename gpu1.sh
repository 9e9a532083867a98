cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
python tools/bench_mvit.py > gpurun_out/r3s/mvit.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3s/stats -- python3 $GRAFT_REPO_ROOT/tools/bench_mvit.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3s/stats.err
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/r3s/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r3s/mvit_kernel_stats.csv
rm -rf gpurun_out/r3s/stats
cat gpurun_out/r3s/mvit.log
