cd $GRAFT_REPO_ROOT
# usage: tools/gpu_prof.sh [round-tag, default r05] [outdir-name, default <tag>p]
T=${1:-r05}
D=${2:-${T}p}
mkdir -p gpurun_out/$D
bash tools/profile_round.sh $D $T fp32 vis pmc > gpurun_out/$D/log_fp32.txt 2>&1
bash tools/profile_round.sh $D $T fp32 av > gpurun_out/$D/log_fp32_av.txt 2>&1
bash tools/profile_round.sh $D $T bf16 vis pmc > gpurun_out/$D/log_bf16.txt 2>&1
bash tools/profile_round.sh $D $T fp16 vis > gpurun_out/$D/log_fp16.txt 2>&1
bash tools/profile_train.sh $D $T > gpurun_out/$D/log_train.txt 2>&1
cd $GRAFT_REPO_ROOT
# the bench lines below read the same-build PMC / trace artefacts from profiles/ (roofline.traffic, dominant_kernel.mfma_busy)
cp gpurun_out/$D/${T}_hbm_traffic_*.json gpurun_out/$D/${T}_*_kernel_stats.csv gpurun_out/$D/${T}_pmc_mfma_busy_*.md gpurun_out/$D/${T}_manifest.jsonl profiles/
python3 bench.py --steps 50 --warmup 10 > gpurun_out/$D/${T}_bench.json 2> gpurun_out/$D/bench.err
python3 bench.py --steps 50 --warmup 10 --mode av --no-cpu-baseline > gpurun_out/$D/${T}_bench_av.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 50 --warmup 10 --precision bf16 --no-cpu-baseline --dump-launches gpurun_out/$D/${T}_launches_bf16_unprofiled.json > gpurun_out/$D/${T}_bench_bf16.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 20 --warmup 5 --precision fp16 --batch 64 --mode av --no-cpu-baseline --no-train-leg > gpurun_out/$D/${T}_bench_fp16_b64_av.json 2>> gpurun_out/$D/bench.err
bash tools/batch_sweep.sh gpurun_out/$D/${T}_batch_sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_salunet.py -q -s 2>&1 | grep -i "err\|diff\|max\|passed" | head -60 > gpurun_out/$D/parity_prints.txt
rm -rf gpurun_out/$D/stats_* gpurun_out/$D/pmc_fetch_* gpurun_out/$D/pmc_write_* gpurun_out/$D/pmc_busy_*/
ls gpurun_out/$D | head -60
