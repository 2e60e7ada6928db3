cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -x -q -k "occ_mlp" 2>&1 | tail -15
