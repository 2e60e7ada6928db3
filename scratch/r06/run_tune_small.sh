# Extends the shipped TunableOp table with the GEMM shapes of the lifting step at 1 and 8 viewpoints per step (config.latency:
# the reference's own operating point, vocc.py:222): untuned, the library runs them on its default heuristics (stream-K
# fallbacks, 12-workgroup launches for the 450-row products).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
cp vln-ver_amd/tuning/tunableop_gfx950_vocc.csv gpurun_out/r06/tunableop_small0.csv   # TunableOp appends the device ordinal
for B in 1 8; do python3 scratch/r06/small_step.py $B 20 1; done
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/gpurun_out/r06/tunableop_small.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=10
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3
T0=$(date +%s)
for B in 1 8; do
  PYTORCH_TUNABLEOP_TUNING=1 timeout 1500 python3 scratch/r06/small_step.py $B 2 0 2> gpurun_out/r06/tune_b$B.err
  echo "tune B=$B rc $? $(( $(date +%s) - T0 )) s; lines $(wc -l < gpurun_out/r06/tunableop_small0.csv) (was $(wc -l < vln-ver_amd/tuning/tunableop_gfx950_vocc.csv))"
done
unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_FILENAME
cp gpurun_out/r06/tunableop_small0.csv vln-ver_amd/tuning/tunableop_gfx950_vocc.csv
for B in 1 8; do python3 scratch/r06/small_step.py $B 20 1; done
