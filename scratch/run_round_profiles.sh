cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -x -q -k "bf16_value" 2>&1 | tail -2
timeout 900 python bench.py --steps 4 --warmup 2 > gpurun_out/r01/bench_default.json 2> gpurun_out/r01/bench_default.err
echo "bench done $?"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/trace -o trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01/trace_bench.json 2> gpurun_out/r01/trace.err
echo "trace done $?"
python scratch/prof_summary.py kernels gpurun_out/r01/trace/trace_results.db gpurun_out/r01/bench_kernel_stats.csv; rm -rf gpurun_out/r01/trace
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r01/pmc_$C -o pmc -- python3 scratch/bench_gather.py 64 > gpurun_out/r01/pmc_$C.log 2>&1
  python scratch/prof_summary.py pmc gpurun_out/r01/pmc_$C/pmc_results.db gpurun_out/r01/gather_pmc.csv
  python scratch/prof_summary.py kernels gpurun_out/r01/pmc_$C/pmc_results.db gpurun_out/r01/gather_kernel_stats_$C.csv
  rm -rf gpurun_out/r01/pmc_$C
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r01/pmcb_$C -o pmc -- python3 scratch/bench_gather.py 64 4x15x15 bf16 > gpurun_out/r01/pmcb_$C.log 2>&1
  python scratch/prof_summary.py pmc gpurun_out/r01/pmcb_$C/pmc_results.db gpurun_out/r01/gather_bf16_pmc.csv
  rm -rf gpurun_out/r01/pmcb_$C
done
echo "pmc done"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/gtraceb -o trace -- python3 scratch/bench_gather.py 64 4x15x15 bf16 > gpurun_out/r01/gtraceb.log 2>&1
python scratch/prof_summary.py kernels gpurun_out/r01/gtraceb/trace_results.db gpurun_out/r01/gather_bf16_kernel_stats.csv; rm -rf gpurun_out/r01/gtraceb
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/gtrace -o trace -- python3 scratch/bench_gather.py 64 > gpurun_out/r01/gtrace.log 2>&1
python scratch/prof_summary.py kernels gpurun_out/r01/gtrace/trace_results.db gpurun_out/r01/gather_kernel_stats.csv; rm -rf gpurun_out/r01/gtrace
ls -la gpurun_out/r01
