# Round-4 evidence for bench.py: kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench
# command itself (header: viewpoints per launch + sha256 of csrc/ver_sca.hip, which bench.py checks before quoting
# roofline.traffic), the MFMA-busy pass, then the bench line.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/r04; mkdir -p $R
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace -o trace -- python3 $CMD > $R/trace_bench.json 2> $R/trace.err; echo "trace $?"
python scratch/prof_summary.py kernels $R/trace/trace_results.db $R/r04_bench_kernel_stats.csv; rm -rf $R/trace
rm -f $R/r04_bench_pmc_fetch_write.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $R/pmc_$C -o pmc -- python3 $CMD > $R/pmc_$C.json 2> $R/pmc_$C.err; echo "pmc $C $?"
  python scratch/prof_summary.py pmc $R/pmc_$C/pmc_results.db $R/r04_bench_pmc_fetch_write.csv; rm -rf $R/pmc_$C
done
echo "# viewpoints_per_launch = 192" >> $R/r04_bench_pmc_fetch_write.csv
echo "# ver_sca_sha256 = $(python -c "import bench; print(bench.source_hash())")" >> $R/r04_bench_pmc_fetch_write.csv
echo "# command: rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 $CMD (separate passes); KiB per dispatch" >> $R/r04_bench_pmc_fetch_write.csv
mkdir -p profiles; cp $R/r04_bench_pmc_fetch_write.csv profiles/   # bench.py reads it from profiles/
rm -f $R/r04_bench_pmc_mfma.csv
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/pmc_mfma -o pmc -- python3 $CMD > $R/pmc_mfma.json 2> $R/pmc_mfma.err; echo "pmc mfma $?"
python scratch/r04/mfma_summary.py $R/pmc_mfma/pmc_results.db $R/r04_bench_pmc_mfma.csv; rm -rf $R/pmc_mfma
rm -f $R/r04_gather_sq_counters.csv
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/pmc_sq -o pmc -- python3 $CMD > $R/pmc_sq.json 2> $R/pmc_sq.err; echo "pmc sq $?"
python scratch/prof_summary.py pmc $R/pmc_sq/pmc_results.db $R/r04_gather_sq_counters.csv; rm -rf $R/pmc_sq
FT="bench.py --workload vocc_full_train --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace_ft -o trace -- python3 $FT > $R/trace_ft.json 2> $R/trace_ft.err; echo "trace full_train $?"
python scratch/prof_summary.py kernels $R/trace_ft/trace_results.db $R/r04_full_train_kernel_stats.csv; rm -rf $R/trace_ft
T0=$(date +%s); timeout 900 python bench.py > $R/r04_bench_default.json 2> $R/bench.err; echo "bench $? wall $(( $(date +%s) - T0 )) s (the driver's command: no flags)"
cat $R/r04_bench_default.json | cut -c1-1500
grep "k_sca\|k_zero" $R/r04_bench_pmc_fetch_write.csv; grep "k_sca_fwd" $R/r04_gather_sq_counters.csv; grep "k_sca\|k_zero\|k_occ" $R/r04_bench_kernel_stats.csv; grep "k_msda3d" $R/r04_full_train_kernel_stats.csv
