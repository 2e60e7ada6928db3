# Extends vln-ver_amd/tuning/tunableop_gfx950_vocc.csv with the GEMM shapes of the current bench.py defaults:
# TunableOp loads the existing table (file name + device ordinal), tunes the shapes it does not hold and writes the
# union back; merge gpurun_out/tunableop_next0.csv into the package table afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp vln-ver_amd/tuning/tunableop_gfx950_vocc.csv gpurun_out/tunableop_next0.csv   # TunableOp appends the device ordinal
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunableop_next.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
PYTORCH_TUNABLEOP_TUNING=1 timeout 2400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tuned-gemms ${BENCH_ARGS:-} > gpurun_out/tune.json 2> gpurun_out/tune.err; echo "tune $?"
wc -l vln-ver_amd/tuning/tunableop_gfx950_vocc.csv gpurun_out/tunableop_next*.csv; cut -c1-120 gpurun_out/tune.json; tail -2 gpurun_out/tune.err
