cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r01b/trace -o trace -- python3 bench.py --steps 2 --warmup 1 --batch 16 --no-cpu-baseline > gpurun_out/r01b/trace_bench.json 2> gpurun_out/r01b/trace.err
python scratch/prof_summary.py kernels gpurun_out/r01b/trace/trace_results.db gpurun_out/r01b/bench_kernel_stats.csv; rm -rf gpurun_out/r01b/trace
