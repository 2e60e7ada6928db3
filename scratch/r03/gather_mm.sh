#!/bin/bash
# matrix-core forward gather: parity tests first, then timing against the corner-slot kernel
timeout 600 python -m pytest tests/test_hip_ops_gpu.py -x -q -k "sca_gather or work_lists" 2>&1 | tail -5
for hd in 1 2 4 8; do
  echo "== VER_SCA_MM_HEADS=$hd"
  VER_SCA_MM_HEADS=$hd timeout 300 python scratch/bench_gather.py 192 4x15x15 bf16 | cut -c1-140
  VER_SCA_MM_HEADS=$hd VER_BENCH_RING=1 timeout 300 python scratch/bench_gather.py 192 4x15x15 bf16 | cut -c1-140
done
echo "== VER_SCA_FWD_MM=0"
VER_SCA_FWD_MM=0 VER_BENCH_RING=1 timeout 300 python scratch/bench_gather.py 192 4x15x15 bf16 | cut -c1-140
VER_SCA_MM_HEADS=4 timeout 200 python scratch/r03/timeline_mm.py 192
