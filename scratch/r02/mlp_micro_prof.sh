cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02; rm -rf gpurun_out/r02/mm
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/mm -o t -- python3 scratch/r02/occ_mlp_micro.py > gpurun_out/r02/mm.log 2>&1
python scratch/prof_summary.py kernels $(ls gpurun_out/r02/mm/*results.db | head -1) gpurun_out/r02/mm_stats.csv
rm -rf gpurun_out/r02/mm
grep "k_occ_mlp" gpurun_out/r02/mm_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
