cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
DIFFSAL_EXTRA_HIPCC_FLAGS=-DDIFFSAL_DEV_STAMPS DIFFSAL_BUILD_JOBS=16 python -m diff_sal_amd.build --force > gpurun_out/r3h/build.log 2>&1
python tools/probe_front_stamps.py > gpurun_out/r3h/stamps.log 2>&1
cat gpurun_out/r3h/stamps.log
