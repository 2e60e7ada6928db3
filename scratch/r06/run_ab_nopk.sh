#!/bin/bash
# GPU box: one box, ver_occ_mlp.hip with packed fp32 (scratch/r06/lib_pk.so, the library as it was) vs without (product library)
mkdir -p gpurun_out/r06
Q="--no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
for i in 1 2 3; do
  VER_HIP_LIB=$PWD/scratch/r06/lib_pk.so python bench.py $Q > gpurun_out/r06/ab_pk_$i.json 2> gpurun_out/r06/ab_pk_$i.err
  python bench.py $Q > gpurun_out/r06/ab_nopk_$i.json 2> gpurun_out/r06/ab_nopk_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/ab_pk_*.json')+glob.glob('gpurun_out/r06/ab_nopk_?.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
VER_LIB=scratch/r06/lib_pk.so python scratch/r06/ws_timeline.py 2>&1 | grep rows
python scratch/r06/ws_timeline.py 2>&1 | grep rows
