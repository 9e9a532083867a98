cd $GRAFT_REPO_ROOT
D=gpurun_out/r5m; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $D/tests_all.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.txt 2>&1
tail -4 $D/tests_all.txt; tail -2 $D/smoke.txt
