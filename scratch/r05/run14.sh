cd $GRAFT_REPO_ROOT
VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_nosums_norow.so timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed "s/^/feature team alone, no row sums: /"
VER_HIP_LIB=$GRAFT_REPO_ROOT/scratch/r05/libver_nosums.so timeout 600 python scratch/r05/occ_mlp_time.py 2>&1 | grep rows | sed "s/^/both teams, no row sums on the feature team: /"
