cd $GRAFT_REPO_ROOT
export DIFFSAL_NO_REBUILD=1
for st in 0 2 4 6; do
echo "=== stagger $st"
DIFFSAL_DMA_STAGGER=$st python tools/probe_dma_stamps.py wino 36 56 96 96 96 2 2>&1 | grep -v amdgpu | grep "kernel span\|unit 3\|unit 8"
DIFFSAL_DMA_STAGGER=$st python tools/probe_dma_stamps.py wino 36 28 48 192 192 2 2>&1 | grep -v amdgpu | grep "kernel span\|unit 3"
DIFFSAL_DMA_STAGGER=$st python tools/probe_dma_stamps.py gemm 48384 192 192 2>&1 | grep -v amdgpu | grep "kernel span\|unit 1"
DIFFSAL_DMA_STAGGER=$st python tools/probe_dma_stamps.py gemm 28560 768 864 2>&1 | grep -v amdgpu | grep "kernel span\|unit 3"
done
