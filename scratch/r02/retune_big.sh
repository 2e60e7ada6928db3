# Re-tune the GEMM shapes of the default bench step whose recorded time is > 2 ms with a longer measurement per candidate
# (the table was recorded with 15 ms per candidate: one to three runs of a 5-14 ms GEMM).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
rows=[l.rstrip('\n') for l in open('vln-ver_amd/tuning/tunableop_gfx950_vocc.csv')]
keep=[l for l in rows if l.startswith('Validator') or float(l.split(',')[-1]) <= 2.0]
open('gpurun_out/tunableop_retune0.csv','w').write('\n'.join(keep)+'\n')
print('kept', len(keep), 'of', len(rows))
PY
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunableop_retune.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10
PYTORCH_TUNABLEOP_TUNING=1 timeout 2700 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tuned-gemms > gpurun_out/retune.json 2> gpurun_out/retune.err; echo "tune $?"
wc -l gpurun_out/tunableop_retune0.csv; cut -c1-160 gpurun_out/retune.json; tail -2 gpurun_out/retune.err
