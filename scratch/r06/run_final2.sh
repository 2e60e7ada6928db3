cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
python -m pytest tests/ -x -q -m gpu --timeout 2400 > $R/gpu_suite3.log 2>&1; tail -3 $R/gpu_suite3.log
python bench.py > $R/bench_default_b.json 2> $R/bench_default_b.err; echo "bench rc=$?"; tail -2 $R/bench_default_b.err
bash scratch/r06/run_small_trace.sh 2>&1 | tail -4
