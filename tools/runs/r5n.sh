cd $GRAFT_REPO_ROOT
D=gpurun_out/r5n; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_lowp.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 > $D/tests1.txt
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph --precision bf16"
run() { "$@" 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "bf16 4-scale sum + conv  $(run $B --set mt_tap16_f32=0)" | tee -a $D/ab.txt
echo "bf16 fp32 tap products   $(run $B)" | tee -a $D/ab.txt
done
B2="python3 bench.py --steps 20 --warmup 5 --precision fp16 --batch 64 --mode av --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
echo "fp16 B=64 4-scale sum + conv $(run $B2 --set mt_tap16_f32=0)" | tee -a $D/ab.txt
echo "fp16 B=64 fp32 tap products  $(run $B2)" | tee -a $D/ab.txt
tail -4 $D/tests1.txt
