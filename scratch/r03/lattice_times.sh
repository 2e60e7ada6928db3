#!/bin/bash
# per-kernel times of the lattice data-movement kernels inside the bench step (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03; rm -rf gpurun_out/r03/lt
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/lt -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= > gpurun_out/r03/lt.log 2>&1
python scratch/prof_summary.py kernels $(ls gpurun_out/r03/lt/*results.db | head -1) gpurun_out/r03/lt_stats.csv
rm -rf gpurun_out/r03/lt
grep "k_lattice\|k_run_copy\|k_convt" gpurun_out/r03/lt_stats.csv | awk -F'",' '{split($1,a,"("); print substr(a[1],1,60), $2}'
cut -c1-120 gpurun_out/r03/lt.log | tail -1
