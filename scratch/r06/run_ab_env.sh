#!/bin/bash
# GPU box: default bench (no extras) with and without one environment switch, alternating x3 on one box.   run_ab_env.sh NAME=VALUE
mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
for i in 1 2 3; do
  env $1 python bench.py $Q > gpurun_out/r06/ab_env_off_$i.json 2> gpurun_out/r06/ab_env_off_$i.err
  python bench.py $Q > gpurun_out/r06/ab_env_on_$i.json 2> gpurun_out/r06/ab_env_on_$i.err
done
python - "$1" <<'PY'
import json,glob,sys
for f in sorted(glob.glob('gpurun_out/r06/ab_env_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        o={k['kernel']:k for k in d['roofline_other_kernels']}
        print(f.replace('off', sys.argv[1]).replace('_on_', '_default_'), d['ms_per_step'], d['value'], 'dgrad ms', o['head_gemm_dgrad']['ms_per_step'], o['head_gemm_dgrad']['achieved'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
