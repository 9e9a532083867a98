cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r3q/t_all.log 2>&1
echo "rc=$?" >> gpurun_out/r3q/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3q/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r3q/smoke.log
tail -n 5 gpurun_out/r3q/t_all.log; tail -n 3 gpurun_out/r3q/smoke.log
