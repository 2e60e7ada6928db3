cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r1 -- python3 bench.py --steps 2 --warmup 1 --batch 16 --micro 8 --dtype bf16 --no-cpu-baseline > gpurun_out/prof1.log 2>&1
ls -R gpurun_out/prof1 | head -30
