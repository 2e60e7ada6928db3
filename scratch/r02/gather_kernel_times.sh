# per-kernel times of the fused gather launches (rocprofv3 kernel trace of scratch/bench_gather.py); usage: <B> <dtype>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
rm -rf gpurun_out/r02/gt
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/gt -o t -- python3 scratch/bench_gather.py ${1:-192} 4x15x15 ${2:-bf16} > gpurun_out/r02/gt.log 2>&1
python scratch/prof_summary.py kernels $(ls gpurun_out/r02/gt/*results.db | head -1) gpurun_out/r02/gather_kernel_stats_${2:-bf16}.csv
rm -rf gpurun_out/r02/gt
grep "k_sca\|k_zero" gpurun_out/r02/gather_kernel_stats_${2:-bf16}.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('%-60s calls %4s avg %9.1f us' % (r[0][:60], r[1], float(r[3])/1e3))"
grep fwd_us gpurun_out/r02/gt.log
