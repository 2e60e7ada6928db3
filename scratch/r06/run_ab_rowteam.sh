#!/bin/bash
# GPU box: one box, the library before the row-team rewrite of k_occ_mlp_bwd_ws (scratch/r06/lib_before_rowteam.so) vs the product library
mkdir -p gpurun_out/r06
python -m pytest tests/ -q -m gpu --timeout 1500 > gpurun_out/r06/gpu_suite_rowteam.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/gpu_suite_rowteam.log; tail -4 gpurun_out/r06/gpu_suite_rowteam.log
Q="--no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
for i in 1 2 3; do
  VER_HIP_LIB=$PWD/scratch/r06/lib_before_rowteam.so python bench.py $Q > gpurun_out/r06/ab_rt_old_$i.json 2> gpurun_out/r06/ab_rt_old_$i.err
  python bench.py $Q > gpurun_out/r06/ab_rt_new_$i.json 2> gpurun_out/r06/ab_rt_new_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/ab_rt_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
