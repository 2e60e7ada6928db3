cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/tq; mkdir -p $R
CMD="bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace -o trace -- python3 $CMD > $R/trace_bench.json 2> $R/trace.err; echo "trace $?"
python scratch/prof_summary.py kernels_timed $R/trace/trace_results.db $R/kernel_stats.csv 6 3; rm -rf $R/trace
