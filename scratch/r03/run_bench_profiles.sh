# Round-3 evidence for bench.py: kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench
# command itself, then the bench line (which reads the PMC summary for roofline.traffic).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/r03; mkdir -p $R
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace -o trace -- python3 $CMD > $R/trace_bench.json 2> $R/trace.err; echo "trace $?"
python scratch/prof_summary.py kernels $R/trace/trace_results.db $R/r03_bench_kernel_stats.csv; rm -rf $R/trace
rm -f $R/r03_bench_pmc_fetch_write.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $R/pmc_$C -o pmc -- python3 $CMD > $R/pmc_$C.json 2> $R/pmc_$C.err; echo "pmc $C $?"
  python scratch/prof_summary.py pmc $R/pmc_$C/pmc_results.db $R/r03_bench_pmc_fetch_write.csv; rm -rf $R/pmc_$C
done
echo "# viewpoints_per_launch = 192" >> $R/r03_bench_pmc_fetch_write.csv
echo "# command: rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 $CMD (separate passes); KiB per dispatch" >> $R/r03_bench_pmc_fetch_write.csv
mkdir -p profiles; cp $R/r03_bench_pmc_fetch_write.csv profiles/   # bench.py reads it from profiles/
timeout 900 python bench.py --steps 6 --warmup 2 > $R/r03_bench_default.json 2> $R/bench.err; echo "bench $?"
cat $R/r03_bench_default.json | cut -c1-1500
grep "k_sca\|k_zero" $R/r03_bench_pmc_fetch_write.csv; grep "k_sca\|k_zero" $R/r03_bench_kernel_stats.csv
