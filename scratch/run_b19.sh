cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for B in 32 64; do
for T in 1024 512; do
  for W in 768 1536 3072; do
    echo -n "B=$B bwd threads=$T min_wgs=$W: "
    VER_SCA_BWD_MIN_WGS=$W VER_SCA_BWD_THREADS=$T timeout 120 python scratch/bench_gather.py $B 2>/dev/null | tail -1 | cut -c130-200
  done
done
done
VER_SCA_BWD_THREADS=512 VER_SCA_BWD_MIN_WGS=1536 timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -x -q -k "sca_gather" 2>&1 | tail -3
