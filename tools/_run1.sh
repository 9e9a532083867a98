cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_step.py -q -m gpu 2>&1 | tail -n 4
timeout 1500 python3 bench.py --workload train --mode av --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])
for c in d['roofline']['classes'][:4]: print(c['class'], c['launches'], c['ms'], c['tflops'])"
