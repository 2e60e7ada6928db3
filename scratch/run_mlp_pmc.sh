cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
rm -f gpurun_out/r01/mlp_pmc.csv
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r01/mp_$N -o pmc -- python3 scratch/dbg_mlp.py > gpurun_out/r01/mp_$N.log 2>&1
  python scratch/prof_summary.py pmc gpurun_out/r01/mp_$N/pmc_results.db gpurun_out/r01/mlp_pmc.csv
  python scratch/prof_summary.py kernels gpurun_out/r01/mp_$N/pmc_results.db gpurun_out/r01/mlp_kernel_stats.csv
  rm -rf gpurun_out/r01/mp_$N
done
grep "k_occ_mlp" gpurun_out/r01/mlp_pmc.csv; grep "k_occ_mlp" gpurun_out/r01/mlp_kernel_stats.csv | cut -c1-150
