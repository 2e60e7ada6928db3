#!/bin/bash
for n in main NC main NC; do
  lib=vln-ver_amd/libver_hip.so; [ $n != main ] && lib=scratch/r04/lib_$n.so
  echo -n "$n : "
  VER_LIB=$PWD/$lib VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16 2>&1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['fwd_us'], d['bwd_us'])"
done
