#!/bin/bash
# in-bench effect of the reverse unit order of the gather kernels (memory-side cache) and of the head-major layout
A="--steps 4 --warmup 1 --no-cpu-baseline --sub-records= --latency-batches= --host-fed-steps 0"
for e in "X=0" "VER_SCA_CS_REVERSE=0 VER_SCA_BWD_REVERSE=0" "VER_SCA_HEAD_MAJOR=1" "X=0"; do
  echo -n "$e : "
  env $e python bench.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'bwd', d['roofline_other_kernels'][0]['frac'], d['roofline_other_kernels'][0]['avg_launch_us'])"
done
