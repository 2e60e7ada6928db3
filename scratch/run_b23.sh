cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scratch/prof_full.py 2>&1 | tail -45 | cut -c1-200
