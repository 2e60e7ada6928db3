cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python scratch/r05/full_prof.py --batch 64 --out gpurun_out/full_prof64.txt > gpurun_out/fp64.log 2>&1; echo "prof $?"
tail -3 gpurun_out/fp64.log
