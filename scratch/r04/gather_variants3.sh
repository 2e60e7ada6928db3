#!/bin/bash
for n in M1 M2 M3; do
  lib=scratch/r04/lib_$n.so
  echo -n "$n : "
  VER_LIB=$PWD/$lib VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16 2>&1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['fwd_us'], d['bwd_us'])"
done
python scratch/r04/gemm_ledger.py --out gpurun_out/r04_gemm_ledger.csv > gpurun_out/r04_gemm_ledger.log 2>&1
