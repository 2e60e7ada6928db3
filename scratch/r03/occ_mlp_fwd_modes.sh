#!/bin/bash
# forward occupancy MLP: rows per block / prefetch variants (VER_OCC_MLP_FWD=<RT><PF>), kernel time at 32 M rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for m in ${MODES:-40 41 21 20}; do
  rm -rf gpurun_out/r03/mlp_kt
  VER_OCC_MLP_FWD=$m rocprofv3 --kernel-trace --stats -d gpurun_out/r03/mlp_kt -o t -- python3 scratch/r02/occ_mlp_micro.py 32256000 > gpurun_out/r03/mlp_micro.log 2>&1
  python scratch/prof_summary.py kernels gpurun_out/r03/mlp_kt/t_results.db gpurun_out/r03/mlp_fwd_$m.csv
  echo "== VER_OCC_MLP_FWD=$m"; grep -i "occ_mlp_fwd" gpurun_out/r03/mlp_fwd_$m.csv | awk -F'",' '{split($1,a,"("); print substr(a[1],1,60), $2}'
done
rm -rf gpurun_out/r03/mlp_kt
