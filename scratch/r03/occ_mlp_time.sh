#!/bin/bash
# kernel times of the fused occupancy MLP at 32 M rows (folded first Linear): rocprofv3 kernel trace of the micro-benchmark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for f in ${MODES:-1 0}; do
  rm -rf gpurun_out/r03/mlp_kt
  VER_OCC_MLP_BWD_FUSED=$f rocprofv3 --kernel-trace --stats -d gpurun_out/r03/mlp_kt -o t -- python3 scratch/r02/occ_mlp_micro.py 32256000 > gpurun_out/r03/mlp_micro.log 2>&1
  python scratch/prof_summary.py kernels gpurun_out/r03/mlp_kt/t_results.db gpurun_out/r03/mlp_stats_$f.csv
  echo "== VER_OCC_MLP_BWD_FUSED=$f"; grep -i "occ_mlp\|MT128" gpurun_out/r03/mlp_stats_$f.csv | awk -F'",' '{split($1,a,"("); print substr(a[1],1,50), $2}'
done
rm -rf gpurun_out/r03/mlp_kt
