#!/bin/bash
# dataflow gather: parity (launch-mode test) and timing against the shipped kernel; the pre-store wait on both
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -k "launch_modes" 2>&1 | tail -5
for lib in "" $PWD/scratch/r04/libver_prestore.so; do
for B in 64 128; do
for cfg in "0 2 256" "1 2 256" "1 1 256"; do
  set -- $cfg
  echo "LIB=$(basename "$lib") B=$B DF=$1 CG=$2 WGS=$3"
  VER_LIB=$lib VER_SCA_FWD_DF=$1 VER_SCA_DF_CG=$2 VER_SCA_DF_WGS=$3 VER_BENCH_PREZERO=1 timeout 300 python scratch/bench_gather.py $B 4x15x15 bf16 2>&1 | grep -v amdgpu.ids
done; done; done
