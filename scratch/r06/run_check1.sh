cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
python -m pytest tests/test_hip_ops_gpu.py -q -x -k "class_stacked or upsample_on_gpu or convt_weight or wgrad_tn or clip_adamw" > $R/check1_ops.log 2>&1; tail -3 $R/check1_ops.log
python -m pytest tests/test_head_gpu.py -q -x > $R/check1_head.log 2>&1; tail -3 $R/check1_head.log
python -m pytest tests/test_z_bench_launch.py -q -x -k "two_ranks" > $R/check1_ddp.log 2>&1; tail -3 $R/check1_ddp.log
for B in 1 8; do python3 scratch/r06/small_step.py $B 20 1 2>&1 | tail -1; done
