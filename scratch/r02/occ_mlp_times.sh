# per-kernel times of the occupancy MLP kernels inside the bench step (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02; rm -rf gpurun_out/r02/mt
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/mt -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/mt.log 2>&1
python scratch/prof_summary.py kernels $(ls gpurun_out/r02/mt/*results.db | head -1) gpurun_out/r02/mt_stats.csv
rm -rf gpurun_out/r02/mt
grep "k_lattice_transpose" gpurun_out/r02/mt_stats.csv | cut -c1-70,150-400
cut -c1-200 gpurun_out/r02/mt.log | tail -1
