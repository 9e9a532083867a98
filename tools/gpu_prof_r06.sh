cd $GRAFT_REPO_ROOT
# Round-6 evidence set, one build: the default line (fp32 headline, configs[1] workload), the 16-bit configurations of BASELINE
# (configs[1] as written: bf16 B = 4; configs[4]: fp16, 64 AV clips per pass) each with rocprofv3 kernel stats, launch tables and the
# FETCH_SIZE / WRITE_SIZE / matrix-pipe PMC passes, AV fp32, the training step, batch sweep, parity prints.
# usage: tools/gpu_prof_r06.sh [outdir-name, default r06p]
T=r06
D=${1:-${T}p}
mkdir -p gpurun_out/$D
bash tools/profile_round.sh $D $T fp32 vis pmc > gpurun_out/$D/log_fp32.txt 2>&1
bash tools/profile_round.sh $D $T bf16 vis pmc > gpurun_out/$D/log_bf16.txt 2>&1
BATCH=64 STEPS=20 PSTEPS=3 bash tools/profile_round.sh $D $T fp16 av pmc > gpurun_out/$D/log_fp16_b64_av.txt 2>&1
BATCH=64 STEPS=20 PSTEPS=3 bash tools/profile_round.sh $D $T bf16 av > gpurun_out/$D/log_bf16_b64_av.txt 2>&1
bash tools/profile_train.sh $D $T > gpurun_out/$D/log_train.txt 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/$D/${T}_hbm_traffic_*.json gpurun_out/$D/${T}_*_kernel_stats.csv gpurun_out/$D/${T}_pmc_mfma_busy_*.md gpurun_out/$D/${T}_manifest.jsonl profiles/
python3 bench.py --steps 50 --warmup 10 > gpurun_out/$D/${T}_bench.json 2> gpurun_out/$D/bench.err
python3 bench.py --steps 50 --warmup 10 --mode av --no-cpu-baseline > gpurun_out/$D/${T}_bench_av.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 50 --warmup 10 --precision bf16 --no-cpu-baseline --no-train-leg > gpurun_out/$D/${T}_bench_bf16.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 20 --warmup 5 --precision fp16 --batch 64 --mode av --no-cpu-baseline --no-train-leg > gpurun_out/$D/${T}_bench_fp16_b64_av.json 2>> gpurun_out/$D/bench.err
python3 bench.py --steps 20 --warmup 5 --precision bf16 --batch 64 --mode av --no-cpu-baseline --no-train-leg > gpurun_out/$D/${T}_bench_bf16_b64_av.json 2>> gpurun_out/$D/bench.err
# same box, same build: the 8-byte forms and the kernels the planner took before this round (DIFFSAL_NO_STREAM16=1)
DIFFSAL_NO_STREAM16=1 python3 bench.py --steps 20 --warmup 5 --precision fp16 --batch 64 --mode av --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision > gpurun_out/$D/${T}_bench_fp16_b64_av_round5_forms.json 2>> gpurun_out/$D/bench.err
bash tools/batch_sweep.sh gpurun_out/$D/${T}_batch_sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_salunet.py -q -s 2>&1 | grep -i "err\|diff\|max\|passed" | head -60 > gpurun_out/$D/${T}_parity_prints.txt
rm -rf gpurun_out/$D/stats_* gpurun_out/$D/pmc_fetch_* gpurun_out/$D/pmc_write_* gpurun_out/$D/pmc_busy_*/
ls gpurun_out/$D | head -80
