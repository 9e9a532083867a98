cd $GRAFT_REPO_ROOT
D=r3p
mkdir -p gpurun_out/$D
python3 bench.py --steps 50 --warmup 10 > gpurun_out/$D/r03_bench.json 2> gpurun_out/$D/bench.err
python3 bench.py --steps 50 --warmup 10 --mode av --no-cpu-baseline > gpurun_out/$D/r03_bench_av.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 50 --warmup 10 --precision bf16 --no-cpu-baseline --dump-launches gpurun_out/$D/r03_launches_bf16_unprofiled.json > gpurun_out/$D/r03_bench_bf16.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 20 --warmup 5 --precision fp16 --batch 64 --mode av --no-cpu-baseline --no-train-leg > gpurun_out/$D/r03_bench_fp16_b64_av.json 2>> gpurun_out/$D/bench.err
python tools/bench_mvit.py > gpurun_out/$D/mvit.log 2>&1
for f in r03_bench.json r03_bench_av.json r03_bench_bf16.json r03_bench_fp16_b64_av.json; do head -c 300 gpurun_out/$D/$f | cut -c 1-260; echo; done; cat gpurun_out/$D/mvit.log
