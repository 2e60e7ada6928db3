cd $GRAFT_REPO_ROOT
for RW in 1 0; do VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_nofeat.so VER_OCC_MLP_ROWS4=$RW timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed 's/^/row team alone: /'; done
VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_norow.so timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed 's/^/feature team alone: /'
