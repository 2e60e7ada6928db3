cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
rm -f gpurun_out/r01/gather_sq_pmc.csv
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r01/gp_$N -o pmc -- python3 scratch/bench_gather.py 64 > gpurun_out/r01/gp.log 2>&1
  python scratch/prof_summary.py pmc gpurun_out/r01/gp_$N/pmc_results.db gpurun_out/r01/gather_sq_pmc.csv
  rm -rf gpurun_out/r01/gp_$N
done
grep "k_sca" gpurun_out/r01/gather_sq_pmc.csv
