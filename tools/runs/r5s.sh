cd $GRAFT_REPO_ROOT
D=gpurun_out/r5s; mkdir -p $D
B="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-encoders --no-alt-precision --no-reference-graph"
run() { "$@" 2>>$D/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "default        $(run $B)" | tee -a $D/ab.txt
echo "CONV_DMA=1     $(DIFFSAL_CONV_DMA=1 run $B)" | tee -a $D/ab.txt
echo "CONV_DMA=2     $(DIFFSAL_CONV_DMA=2 run $B)" | tee -a $D/ab.txt
done
DIFFSAL_CONV_DMA=1 $B --dump-launches $D/launches_dma1.json > /dev/null 2>>$D/err.txt
python3 -c "
import json
for l in json.load(open('$D/launches_dma1.json'))['launches']:
    if l['class'] in ('K5',): print(l['class'], l['us'], l['op'][:40], l['kernel'][:70])"
