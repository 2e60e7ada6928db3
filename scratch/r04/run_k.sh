#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/r04k; mkdir -p $R
timeout 1500 python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py -x -q -k "occ_mlp or fused or head" 2>&1 | tail -2
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
timeout 900 rocprofv3 --kernel-trace --stats -d $R/trace -o trace -- python3 $CMD > $R/trace_bench.json 2> $R/trace.err; echo "trace $?"
python scratch/prof_summary.py kernels $R/trace/trace_results.db $R/kernel_stats.csv; rm -rf $R/trace
grep "k_occ_mlp" $R/kernel_stats.csv | cut -c1-60,120-260
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
