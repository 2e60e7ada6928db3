cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "occ_mlp" 2>&1 | tail -4
for RW in 1 0; do VER_OCC_MLP_ROWS4=$RW timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows; done
VER_OCC_MLP_SAVE_RSTD=0 timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows
