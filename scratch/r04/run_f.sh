#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest5.txt 2>&1; tail -3 gpurun_out/r04_pytest5.txt
python -m pytest tests/test_head_gpu.py -m gpu -q -s -k test_vocc_head_bf16_autocast_within_1e2 2>&1 | grep "bf16 head"
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-fed-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['latency'], d['config']['full_train'], d['config']['fp32'])"
python scratch/r04/gemm_ledger.py --out gpurun_out/r04_gemm_ledger_b.csv > /dev/null 2>&1; grep "other-by-op" gpurun_out/r04_gemm_ledger_b.csv | head -40 | cut -c1-190
timeout 900 python scratch/r04/occproj_shapes.py 2>&1 | grep "^{"
