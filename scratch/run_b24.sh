cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_head_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload vocc_full_train --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b24_full32.json 2> gpurun_out/b24_full32.err; echo "full32 $?"
cut -c1-220 gpurun_out/b24_full32.json; tail -2 gpurun_out/b24_full32.err
