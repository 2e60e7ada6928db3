#!/bin/bash
python -m pytest tests/test_head_gpu.py tests/test_hip_ops_gpu.py -m gpu -x -q -k "lattice or upsample or head or vocc" > gpurun_out/r04_pytest9.txt 2>&1; tail -3 gpurun_out/r04_pytest9.txt
for i in 1 2; do
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --host-fed-steps 0 --sub-records= --latency-batches= 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'bwd', d['roofline_other_kernels'][0]['avg_launch_us'])"
done
