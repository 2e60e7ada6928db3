cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 900 python scratch/r05/dgrad_bench.py check big > $R/dgrad1.txt 2>&1; echo "rc $?"
cat $R/dgrad1.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py -x -q -m gpu -k "wgrad or gemm_nn or dgrad or head" 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --sub-records= --latency-batches= --host-fed-steps 0 > $R/bench7.json 2> $R/bench7.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench7.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for o in d['roofline_other_kernels']: print(' ', o['kernel'], o.get('achieved'), o.get('frac'), o.get('ms_per_step'))
PY
tail -2 $R/bench7.err
