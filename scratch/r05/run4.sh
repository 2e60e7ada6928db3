cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
T0=$(date +%s); timeout 900 python bench.py > $R/bench4.json 2> $R/bench4.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench4.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'])
for o in d['roofline_other_kernels']: print(' ', o['kernel'], o.get('achieved'), o.get('frac'), o.get('ms_per_step'), o.get('avg_launch_us'))
c=d['config']; print('latency', c['latency']); print('full_train', c['full_train']); print('fp32', c['fp32']); print('host_fed', c['host_fed']); print('cpu', d['cpu_baseline']['value'])
PY
tail -3 $R/bench4.err
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
