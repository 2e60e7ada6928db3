cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python scratch/exp_wgrad.py 2>&1 | tail -8
