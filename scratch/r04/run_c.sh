#!/bin/bash
# (1) GPU tests of the latest changes, (2) kernel traces of the step at the reference's batch points (B = 8, B = 1)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py -m gpu -x -q -k "occ_mlp or head_major or vocc_head or sca" > gpurun_out/r04_pytest3.txt 2>&1; tail -3 gpurun_out/r04_pytest3.txt
R=gpurun_out/r04; mkdir -p $R
for B in 8 1; do
  CMD="bench.py --batch $B --micro $B --steps 10 --warmup 3 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/trace_b$B -o trace -- python3 $CMD > $R/trace_b$B.json 2> $R/trace_b$B.err; echo "trace B=$B $?"
  python scratch/prof_summary.py kernels $R/trace_b$B/trace_results.db $R/r04_b${B}_kernel_stats.csv; rm -rf $R/trace_b$B
  python bench.py --batch $B --micro $B --steps 20 --warmup 5 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records= 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('B=$B', d['value'], d['ms_per_step'])"
done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sub-records= --latency-batches= --host-fed-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'])"
