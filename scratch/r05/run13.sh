cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "occ_mlp" 2>&1 | tail -2
timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed 's/^/SB0: /'
for L in 1 2; do VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_sb$L.so timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed "s/^/SB$L: /"; done
VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_sb2.so timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "occ_mlp" 2>&1 | tail -2
