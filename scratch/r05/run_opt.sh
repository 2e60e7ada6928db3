cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/opt; mkdir -p $R
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -k "clip_adamw" > $R/tests.log 2>&1; echo "tests $?"; tail -6 $R/tests.log
for m in "--torch-optimizer" ""; do
timeout 600 python bench.py --no-cpu-baseline --sub-records= --host-fed-steps 0 $m > $R/bench.json 2> $R/bench.err; echo "bench $?"; tail -2 $R/bench.err
python -c "
import json;d=json.loads(open('$R/bench.json').read().strip().splitlines()[-1]);print('line [$m]', d['value'], d['ms_per_step'], [l['ms_per_step'] for l in d['config'].get('latency')], d['config']['optimizer'])"
done
