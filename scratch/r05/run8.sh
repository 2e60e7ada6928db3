cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 1500 python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -4
T0=$(date +%s); timeout 900 python bench.py > $R/bench8.json 2> $R/bench8.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench8.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'])
for o in d['roofline_other_kernels']: print(' ', o['kernel'], o.get('achieved'), o.get('frac'), o.get('ms_per_step'))
c=d['config']; print('latency', [(l['viewpoints_per_gpu_per_step'], l['ms_per_step']) for l in c['latency']]); print('full_train', c['full_train']['ms_per_step'], c['full_train']['viewpoints_per_s']); print('fp32', c['fp32']['ms_per_step'])
PY
