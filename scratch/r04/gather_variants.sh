#!/bin/bash
# round 4: prologue experiments on k_sca_fwd_cs (DMA first / pipelined conversion / prologue priority), timing per variant
for n in main B C D E main E; do
  lib=vln-ver_amd/libver_hip.so; [ $n != main ] && lib=scratch/r04/lib_$n.so
  echo -n "$n : "
  VER_LIB=$PWD/$lib VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16 2>&1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['fwd_us'], d['bwd_us'])"
done
VER_LIB=$PWD/scratch/r04/lib_T.so python scratch/r04/timeline_cs.py 2>&1 | head -16
