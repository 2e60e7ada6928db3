#!/bin/bash
# kernel times of the fused occupancy MLP at 32 M rows (folded first Linear): rocprofv3 kernel trace of the micro-benchmark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03/mlp_kt
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/mlp_kt -o t -- python3 scratch/r02/occ_mlp_micro.py 32256000 > gpurun_out/r03/mlp_micro.log 2>&1
tail -2 gpurun_out/r03/mlp_micro.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03/mlp_kt/**/*kernel_stats.csv', recursive=True)
for row in csv.DictReader(open(f[0])):
    if 'occ_mlp' in row['Name'] or float(row['Percentage']) > 5:
        print('%-60s calls %s avg %.3f ms' % (row['Name'][:60], row['Calls'], float(row['AverageNs']) / 1e6))
PY
