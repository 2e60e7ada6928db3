# Round-5 final check of the tree: GPU suite, smoke(), then the profiles of the bench command (scratch/r05/run_bench_profiles.sh).
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
bash scratch/r05/run_bench_profiles.sh
