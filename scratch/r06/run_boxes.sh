cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/bench_box_$1.json 2> gpurun_out/r06/bench_box_$1.err; echo "rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r06/bench_box_$1.json').read().strip().splitlines()[-1])
r=d['roofline']
print('box $1:', d['value'], d['ms_per_step'], 'roofline', r['frac'], r['avg_launch_us'], 'with zero fill', r.get('frac_with_zero_fill'), (r.get('zero_fill') or {}).get('unhidden_us'), 'latency', [(l['viewpoints_per_gpu_per_step'], l['ms_per_step'], l['graphed']) for l in d['config']['latency']], 'full_train', d['config']['full_train']['viewpoints_per_s'], 'probe', d['cpu_baseline']['rel_diff'])
PY
