cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "occ_mlp" 2>&1 | tail -3
for RW in 1 0; do VER_OCC_MLP_ROWS4=$RW timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows; done
VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_norow.so timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed 's/^/feature team alone: /'
