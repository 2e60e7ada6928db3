# Extends the shipped TunableOp table with the GEMM shapes of the full multi-task step at 64 viewpoints (config.full_train):
# the decoder's Linears over [64 x 900, 768] rows ran the library's stream-K fallback (profiles/r05_full_train_kernel_stats.csv).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp vln-ver_amd/tuning/tunableop_gfx950_vocc.csv gpurun_out/tunableop_full0.csv   # TunableOp appends the device ordinal
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/tunableop_full.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=10
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3
T0=$(date +%s)
PYTORCH_TUNABLEOP_TUNING=1 timeout 2400 python bench.py --workload vocc_full_train --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-tuned-gemms --latency-batches= --host-fed-steps 0 --sub-records= > gpurun_out/tune_full.json 2> gpurun_out/tune_full.err
echo "tune rc $? $(( $(date +%s) - T0 )) s; lines $(wc -l < gpurun_out/tunableop_full0.csv) (was $(wc -l < vln-ver_amd/tuning/tunableop_gfx950_vocc.csv))"
tail -2 gpurun_out/tune_full.err
unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_FILENAME
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 600 python bench.py --workload vocc_full_train --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records= 2>/dev/null | show
cp gpurun_out/tunableop_full0.csv vln-ver_amd/tuning/tunableop_gfx950_vocc.csv
timeout 600 python bench.py --workload vocc_full_train --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records= 2>/dev/null | show
