cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -m gpu -k "occ_mlp" 2>&1 | tail -5
bash scratch/r05/run_bench_profiles.sh
