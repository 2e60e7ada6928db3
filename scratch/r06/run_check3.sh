cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
python -m pytest tests/test_hip_ops_gpu.py -q -x -k "upsample or lattice or convt or class_stacked" > $R/check3_ops.log 2>&1; tail -2 $R/check3_ops.log
python -m pytest tests/test_regime_gpu.py -q -x -s > $R/check3_regime.log 2>&1; grep "B = 192\|passed\|failed" $R/check3_regime.log
python bench.py --steps 4 --warmup 1 --sub-records "" --host-fed-steps 0 --no-cpu-baseline --latency-batches "" > $R/bench_implicit.json 2> $R/bench_implicit.err; tail -2 $R/bench_implicit.err
VER_IMPLICIT_TAPS=0 python bench.py --steps 4 --warmup 1 --sub-records "" --host-fed-steps 0 --no-cpu-baseline --latency-batches "" > $R/bench_explicit.json 2> $R/bench_explicit.err
python - <<'PY'
import json
for n in ('implicit','explicit'):
    d=json.loads(open('gpurun_out/r06/bench_%s.json'%n).read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['config']['peak_hbm_gib'], [(o['kernel'], o.get('ms_per_step'), o['achieved']) for o in d['roofline_other_kernels'] if o['bound']=='mfma'])
PY
