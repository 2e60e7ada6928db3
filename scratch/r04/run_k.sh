#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_ops_gpu.py tests/test_encoder_gpu.py -x -q -k "sca or spatial or encoder" 2>&1 | tail -3
for B in 64 192; do
  VER_BENCH_PREZERO=1 VER_BENCH_RING=1 timeout 300 python scratch/bench_gather.py $B 4x15x15 bf16 2>&1 | grep -v amdgpu.ids
done
timeout 300 python scratch/bench_gather.py 32 4x15x15 2>&1 | grep -v amdgpu.ids
