cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scratch/torch_prof.py --batch 64 --micro 64 --out gpurun_out/torch_prof64.txt > gpurun_out/tp32b.log 2>&1; echo "prof $?"
