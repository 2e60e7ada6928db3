# Round-6 evidence for bench.py (scratch/r04/run_bench_profiles.sh + the verdict's item 1a): the kernel trace is summarised over
# the dispatches of the TIMED steps only (2 priming + 2 warm-up steps dropped: first-touch launches of the gather are the slow
# ones), so that the trace's average of k_sca_fwd_cs can be held against the line's avg_launch_us.
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06p; mkdir -p $R
CMD="bench.py --steps 6 --warmup 2 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace -o trace -- python3 $CMD > $R/trace_bench.json 2> $R/trace.err; echo "trace $?"
python scratch/prof_summary.py kernels $R/trace/trace_results.db $R/r06_bench_kernel_stats_all_steps.csv
python scratch/prof_summary.py kernels_timed $R/trace/trace_results.db $R/r06_bench_kernel_stats.csv 10 4; rm -rf $R/trace
CMD2="bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
rm -f $R/r06_bench_pmc_fetch_write.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $R/pmc_$C -o pmc -- python3 $CMD2 > $R/pmc_$C.json 2> $R/pmc_$C.err; echo "pmc $C $?"
  python scratch/prof_summary.py pmc $R/pmc_$C/pmc_results.db $R/r06_bench_pmc_fetch_write.csv; rm -rf $R/pmc_$C
done
echo "# viewpoints_per_launch = 192" >> $R/r06_bench_pmc_fetch_write.csv
echo "# ver_sca_sha256 = $(python -c "import bench; print(bench.source_hash())")" >> $R/r06_bench_pmc_fetch_write.csv
echo "# command: rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 $CMD2 (separate passes); KiB per dispatch" >> $R/r06_bench_pmc_fetch_write.csv
mkdir -p profiles; cp $R/r06_bench_pmc_fetch_write.csv profiles/   # bench.py reads it from profiles/
rm -f $R/r06_bench_pmc_mfma.csv
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/pmc_mfma -o pmc -- python3 $CMD2 > $R/pmc_mfma.json 2> $R/pmc_mfma.err; echo "pmc mfma $?"
python scratch/r05/mfma_summary.py $R/pmc_mfma/pmc_results.db $R/r06_bench_pmc_mfma.csv > /dev/null; rm -rf $R/pmc_mfma
FT="bench.py --workload vocc_full_train --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace_ft -o trace -- python3 $FT > $R/trace_ft.json 2> $R/trace_ft.err; echo "trace full_train $?"
python scratch/prof_summary.py kernels $R/trace_ft/trace_results.db $R/r06_full_train_kernel_stats.csv; rm -rf $R/trace_ft



T0=$(date +%s); timeout 900 python bench.py > $R/r06_bench_default.json 2> $R/bench.err; echo "bench $? wall $(( $(date +%s) - T0 )) s (the driver's command: no flags)"
cut -c1-600 $R/r06_bench_default.json
grep "k_sca\|k_zero" $R/r06_bench_pmc_fetch_write.csv; grep "k_sca\|k_zero\|k_occ\|k_wgrad" $R/r06_bench_kernel_stats.csv; head -8 $R/r06_bench_pmc_mfma.csv
