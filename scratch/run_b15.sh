cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t15.log 2>&1; echo "tests $?"; tail -8 gpurun_out/t15.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b15.json 2> gpurun_out/b15.err; echo "bench $?"
cut -c1-250 gpurun_out/b15.json; tail -3 gpurun_out/b15.err
