cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunableop.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
export PYTORCH_TUNABLEOP_VERBOSE=0
PYTORCH_TUNABLEOP_TUNING=1 timeout 1500 python bench.py --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/tune_run.json 2> gpurun_out/tune_run.err; echo "tune $?"
ls -la gpurun_out/tunableop*.csv | head; wc -l gpurun_out/tunableop*.csv
cut -c1-200 gpurun_out/tune_run.json; tail -3 gpurun_out/tune_run.err
PYTORCH_TUNABLEOP_TUNING=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/tuned_run.json 2> gpurun_out/tuned_run.err; echo "tuned $?"
cut -c1-200 gpurun_out/tuned_run.json; tail -3 gpurun_out/tuned_run.err
