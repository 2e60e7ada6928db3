cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -x -q -k "sca_gather" 2>&1 | tail -3
for T in 1024 512; do
  for B in 32 64 256; do
    VER_SCA_FWD_THREADS=$T timeout 120 python scratch/bench_gather.py $B 2>/dev/null | tail -1 | cut -c1-230
  done
done
VER_SCA_FWD_THREADS=512 timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -x -q -k "sca_gather" 2>&1 | tail -3
