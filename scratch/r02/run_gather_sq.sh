# SQ counters of the fused gather kernels (bf16 tiles, B viewpoints per launch); usage: run_gather_sq.sh <tag> [B]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; B=${2:-192}
mkdir -p gpurun_out/r02 gpurun_out/r03
OUT=gpurun_out/r02/gather_sq_$TAG.csv
rm -f $OUT
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r02/gp_$N -o pmc -- python3 scratch/bench_gather.py $B 4x15x15 ${DT:-bf16} > gpurun_out/r02/gp_$TAG.log 2>&1
  python scratch/prof_summary.py pmc gpurun_out/r02/gp_$N/pmc_results.db $OUT
  rm -rf gpurun_out/r02/gp_$N
done
grep "k_sca_fwd\|k_zero" $OUT
