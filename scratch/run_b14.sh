cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python scratch/dbg_mlp.py 2>&1 | tail -12
