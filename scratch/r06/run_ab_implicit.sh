cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
for mode in 1 0 3 1 0; do
VER_IMPLICIT_TAPS=$mode python bench.py --steps 5 --warmup 1 --sub-records "" --host-fed-steps 0 --no-cpu-baseline --latency-batches "" > $R/bench_ab_$mode.json 2> $R/bench_ab_$mode.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r06/bench_ab_$mode.json').read().strip().splitlines()[-1])
print('VER_IMPLICIT_TAPS=$mode', d['value'], d['ms_per_step'], d['config']['peak_hbm_gib'], [(o['kernel'], o.get('ms_per_step'), o['achieved']) for o in d['roofline_other_kernels'] if o['bound']=='mfma' and o['kernel'] in ('ver_gemm_nn','ver_wgrad_tn','head_gemms')])
PY
done
