# Extends the shipped TunableOp table with the GEMM shapes of the reference's batch points (8 and 1 viewpoints per step:
# bench.py's config.latency).  Each pass loads the table so far, tunes what it does not hold and writes the union.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp vln-ver_amd/tuning/tunableop_gfx950_vocc.csv gpurun_out/tunableop_lat0.csv   # TunableOp appends the device ordinal
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunableop_lat.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=10
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3
for B in 8 1; do
  T0=$(date +%s)
  PYTORCH_TUNABLEOP_TUNING=1 timeout 1500 python bench.py --batch $B --micro $B --steps 2 --warmup 1 --no-cpu-baseline --no-tuned-gemms --latency-batches= --host-fed-steps 0 --sub-records= > gpurun_out/tune_b$B.json 2> gpurun_out/tune_b$B.err
  echo "tune B=$B rc $? $(( $(date +%s) - T0 )) s; lines $(wc -l < gpurun_out/tunableop_lat0.csv)"
  cp gpurun_out/tunableop_lat0.csv gpurun_out/tunableop_lat_after_b$B.csv
done
tail -2 gpurun_out/tune_b1.err
# before / after at the two batch points
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print([(l['viewpoints_per_gpu_per_step'], l['ms_per_step']) for l in d['config']['latency']])"; }
unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_FILENAME
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --host-fed-steps 0 --sub-records= 2>/dev/null | show
cp gpurun_out/tunableop_lat0.csv vln-ver_amd/tuning/tunableop_gfx950_vocc.csv
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --host-fed-steps 0 --sub-records= 2>/dev/null | show
