#!/bin/bash
# validation + numbers of the round-4 state: gather microbench in the three forms, value_proj forms, GPU tests, bench
for e in "X=0" "VER_BENCH_HM=1" "VER_BENCH_HM=1 VER_BENCH_PREZERO=1"; do
  echo -n "$e : "
  env $e VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16 2>&1 | tail -1
done
python scratch/r04/value_proj_forms.py 2>&1 | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/r04_pytest2.txt 2>&1; tail -3 gpurun_out/r04_pytest2.txt
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench2.json 2> gpurun_out/r04_bench2.err; cut -c1-300 gpurun_out/r04_bench2.json
