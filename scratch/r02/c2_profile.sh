cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02; rm -rf gpurun_out/r02/c2
python bench.py --workload c2_single_scale_fwd --dtype fp32 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-400
python bench.py --workload c2_single_scale_fwd --dtype fp32 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/c2 -o t -- python3 bench.py --workload c2_single_scale_fwd --dtype fp32 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/c2.log 2>&1
python scratch/prof_summary.py kernels $(ls gpurun_out/r02/c2/*results.db | head -1) gpurun_out/r02/c2_stats.csv
rm -rf gpurun_out/r02/c2
head -16 gpurun_out/r02/c2_stats.csv | cut -c1-90,140-260
