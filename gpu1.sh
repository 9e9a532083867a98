cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoders.py tests/test_gpu_encoder_train.py tests/test_abi.py -x -q > $O/t_enc.log 2>&1
echo "rc=$?" >> $O/t_enc.log
python tools/bench_mvit.py > $O/mvit.log 2>&1
python tools/tune_wgrad_mvit.py > $O/sweep_wgrad.log 2>&1
python bench.py --workload train --mode av --steps 10 --warmup 3 --repeats 2 > $O/train.json 2> $O/train.err
tail -n 4 $O/t_enc.log; cat $O/mvit.log; cat $O/sweep_wgrad.log;  head -c 330 $O/train.json
