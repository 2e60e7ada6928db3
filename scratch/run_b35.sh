cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -2
