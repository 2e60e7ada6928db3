cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/ws; mkdir -p $R
timeout 1500 python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py -x -q -k "convt or upsample or lattice or head or occ_proj or vocc or multitask" > $R/tests.log 2>&1; echo "tests $?"; tail -4 $R/tests.log
timeout 600 python bench.py --no-cpu-baseline --sub-records= --host-fed-steps 0 > $R/bench.json 2> $R/bench.err; echo "bench $?"
python -c "
import json;d=json.loads(open('$R/bench.json').read().strip().splitlines()[-1]);print('line', d['value'], d['ms_per_step'], d['config'].get('latency'))"
