cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scratch/torch_prof.py --batch 8 --micro 192 --out gpurun_out/torch_prof8.txt > gpurun_out/tp8.log 2>&1; echo "prof $?"
timeout 300 python scratch/torch_prof.py --batch 1 --micro 192 --out gpurun_out/torch_prof1.txt > gpurun_out/tp1.log 2>&1; echo "prof $?"
