cd $GRAFT_REPO_ROOT
R=gpurun_out/r05p; mkdir -p $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
T0=$(date +%s); timeout 900 python bench.py > $R/r05_bench_default.json 2> $R/bench.err; echo "bench $? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05p/r05_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'])
for o in d['roofline_other_kernels']: print(' ', o['kernel'], o.get('achieved'), o.get('frac'), o.get('ms_per_step'))
c=d['config']; print('latency', [(l['viewpoints_per_gpu_per_step'], l['ms_per_step']) for l in c['latency']]); print('full_train', c['full_train']['ms_per_step'], c['full_train']['viewpoints_per_s'], c['full_train']['graphed']); print('fp32', c['fp32']['viewpoints_per_s'], 'host_fed', c['host_fed']['viewpoints_per_s'], 'cpu', d['cpu_baseline']['value'])
PY
