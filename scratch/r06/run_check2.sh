cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
python -m pytest tests/test_hip_ops_gpu.py -q -x > $R/check2_ops.log 2>&1; tail -3 $R/check2_ops.log
python -m pytest tests/test_head_gpu.py tests/test_encoder_gpu.py tests/test_detector_gpu.py -q -x > $R/check2_head.log 2>&1; tail -3 $R/check2_head.log
for B in 1 8; do python3 scratch/r06/small_step.py $B 20 1 2>&1 | tail -1; done
