cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/rows; mkdir -p $R
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -x -q -k "lattice_rows or fused_rows or lattice or upsample" > $R/tests.log 2>&1; echo "tests $?"; tail -6 $R/tests.log
timeout 900 python -m pytest tests/test_head_gpu.py -x -q > $R/tests2.log 2>&1; echo "tests2 $?"; tail -3 $R/tests2.log
for m in 0 1; do
VER_LATTICE_ROWS=$m timeout 600 python bench.py --no-cpu-baseline --sub-records= --host-fed-steps 0 > $R/bench$m.json 2> $R/bench$m.err; echo "bench $?"
python -c "
import json;d=json.loads(open('$R/bench$m.json').read().strip().splitlines()[-1]);print('line', $m, d['value'], d['ms_per_step'], [l['ms_per_step'] for l in d['config'].get('latency')], d['config']['peak_hbm_gib'])"
done
