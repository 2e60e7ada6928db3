#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest4.txt 2>&1; tail -3 gpurun_out/r04_pytest4.txt
python scratch/r04/gemm_ledger.py --workload vocc_full_train --batch 64 --micro 64 --out gpurun_out/r04_gemm_ledger_full64.csv > /dev/null 2>&1; tail -32 gpurun_out/r04_gemm_ledger_full64.csv | cut -c1-120
for B in 8 1; do
  python bench.py --batch $B --micro $B --steps 20 --warmup 5 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records= 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('B=$B', d['value'], d['ms_per_step'])"
done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['full_train'], d['config']['fp32'])"
