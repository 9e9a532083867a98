cd $GRAFT_REPO_ROOT
D=gpurun_out/r5t; mkdir -p $D
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
run() { "$@" 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "default             $(run $B)" | tee -a $D/ab.txt
echo "side-stream K13     $(run $B --set side_stream_reduce_temp=1)" | tee -a $D/ab.txt
done
echo "bf16 default        $(run $B --precision bf16)" | tee -a $D/ab.txt
echo "bf16 side-stream    $(run $B --precision bf16 --set side_stream_reduce_temp=1)" | tee -a $D/ab.txt
tail -3 $D/err.txt
