#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest8.txt 2>&1; tail -4 gpurun_out/r04_pytest8.txt
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-fed-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'bwd', d['roofline_other_kernels'][0]['frac'], d['roofline_other_kernels'][0]['avg_launch_us'], d['config']['latency'], d['config']['full_train'], d['config']['fp32'])"
