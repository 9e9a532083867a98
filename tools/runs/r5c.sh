cd $GRAFT_REPO_ROOT
D=gpurun_out/r5c; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_gemm_dma.py tests/test_gpu_upconv.py -x -q 2>&1 | tail -15 > $D/tests1.txt
timeout 900 python -m pytest tests/test_gpu_salunet.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15 > $D/tests2.txt
python3 tools/bench_batch_tile.py > $D/batch_tile.txt 2>&1
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph --dump-launches $D/launches_fp32.json > $D/bench_fp32.json 2>$D/bench.err
tail -4 $D/tests1.txt; tail -4 $D/tests2.txt; cat $D/batch_tile.txt; python3 -c "
import json; d=json.loads(open('$D/bench_fp32.json').readline()); print(d['value'], d['ms_per_step'])"
