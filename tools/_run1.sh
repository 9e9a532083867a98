cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/w4/st -- python3 $GRAFT_REPO_ROOT/bench.py --batch 4 --steps 50 --repeats 1 --no-cpu-baseline --no-alt-precision --no-encoders --no-train-leg --no-reference-graph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/w4/st/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:40]:
    n=r['Name']
    if any(k in n for k in ('up2','index','gather','wino4','Index','copy')): print(f"{n[:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
rm -rf gpurun_out/w4/st
