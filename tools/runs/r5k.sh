cd $GRAFT_REPO_ROOT
D=gpurun_out/r5k; mkdir -p $D
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
run() { "$@" 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "default            $(run $B)" | tee -a $D/ab.txt
echo "qkv grouped all    $(run $B --set group_qkv_all=1)" | tee -a $D/ab.txt
echo "qkv grouped all, persistent 512 $(DIFFSAL_GROUP_GRID=512 run $B --set group_qkv_all=1)" | tee -a $D/ab.txt
echo "persistent 512 (K13 too)        $(DIFFSAL_GROUP_GRID=512 run $B)" | tee -a $D/ab.txt
done
