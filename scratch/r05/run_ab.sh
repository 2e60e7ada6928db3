cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/ab; mkdir -p $R
timeout 1200 python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py -x -q -k "focal or head or vocc or multitask or loss" > $R/tests.log 2>&1; echo "tests $?"; tail -3 $R/tests.log
timeout 600 python bench.py --no-cpu-baseline --sub-records= --host-fed-steps 0 > $R/bench.json 2> $R/bench.err; echo "bench $?"; tail -2 $R/bench.err | cut -c1-200
python -c "
import json;d=json.loads(open('$R/bench.json').read().strip().splitlines()[-1]);print('line', d['value'], d['ms_per_step'], [l['ms_per_step'] for l in d['config'].get('latency')])"
