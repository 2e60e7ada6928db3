cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for B in 32 64 256; do
for T in 1024 512; do
  for W in 768 1536 3072; do
    echo -n "B=$B threads=$T min_wgs=$W: "
    VER_SCA_FWD_MIN_WGS=$W VER_SCA_FWD_THREADS=$T timeout 120 python scratch/bench_gather.py $B 2>/dev/null | tail -1 | cut -c60-130
  done
done
done
