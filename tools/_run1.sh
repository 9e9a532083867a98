cd $GRAFT_REPO_ROOT
export DIFFSAL_NO_REBUILD=1
python tools/probe_dma_stamps.py wino 36 56 96 96 96 2 2>&1 | grep -v amdgpu | tail -4
python tools/probe_dma_stamps.py wino 4 14 24 768 768 1 2>&1 | grep -v amdgpu | tail -4
