cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo bench $?
python -c "
import json;d=json.loads(open('gpurun_out/final_bench.json').read().strip().splitlines()[-1]);c=d['config'];print(d['value'], d['ms_per_step'], d['roofline']['frac'], [l['ms_per_step'] for l in c['latency']], c['full_train']['viewpoints_per_s'])"
