#!/bin/bash
# forward gather, fp16 accumulate mode: workgroup size x waves per SIMD (ring offsets, 192 viewpoints per launch)
for wpe in 4 5; do for t in 192 256 320 384 512; do
  echo -n "threads=$t waves_per_simd=$wpe  "
  VER_SCA_FWD_MATH=2 VER_SCA_CS_WAVES_PER_SIMD=$wpe VER_SCA_CS_THREADS_BF16=$t VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fwd_us'])"
done; done
