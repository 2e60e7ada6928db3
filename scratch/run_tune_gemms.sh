cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunableop_b64.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
PYTORCH_TUNABLEOP_TUNING=1 timeout 2000 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-tuned-gemms > gpurun_out/tune64.json 2> gpurun_out/tune64.err; echo "tune $?"
wc -l gpurun_out/tunableop_b64*.csv; cut -c1-200 gpurun_out/tune64.json; tail -2 gpurun_out/tune64.err
