#!/bin/bash
# forward gather at 192 viewpoints per launch: the three tile-math modes, random offsets and the reference's initial ring
for m in 0 1 2; do
  echo "== VER_SCA_FWD_MATH=$m"
  VER_SCA_FWD_MATH=$m python scratch/bench_gather.py 192 4x15x15 bf16
  VER_SCA_FWD_MATH=$m VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16
done
