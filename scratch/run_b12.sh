cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t12.log 2>&1; echo "tests $?"; tail -8 gpurun_out/t12.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b12.json 2> gpurun_out/b12.err; echo "bench $?"
cut -c1-250 gpurun_out/b12.json; tail -3 gpurun_out/b12.err
timeout 300 python scratch/torch_prof.py --batch 32 --micro 32 --out gpurun_out/torch_prof32.txt > gpurun_out/tp32.log 2>&1; echo "prof $?"
