# BASELINE configs[4] (whole vocc.py head: encoder + detection decoder + multi-task heads + losses) on one GPU:
# bench line and rocprofv3 kernel-trace summary
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/r03; mkdir -p $R
timeout 900 python bench.py --workload vocc_full_train --steps 6 --warmup 2 > $R/r03_bench_full_train.json 2> $R/full.err; echo "bench $?"
cut -c1-300 $R/r03_bench_full_train.json
rm -rf $R/trace_full
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace_full -o trace -- python3 bench.py --workload vocc_full_train --steps 2 --warmup 1 --latency-batches= --host-fed-steps 0 > $R/trace_full.json 2> $R/trace_full.err; echo "trace $?"
python scratch/prof_summary.py kernels $R/trace_full/trace_results.db $R/r03_full_train_kernel_stats.csv; rm -rf $R/trace_full
head -12 $R/r03_full_train_kernel_stats.csv | cut -c1-150
