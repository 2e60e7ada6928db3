cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t27.log 2>&1; echo "tests $?"; tail -4 gpurun_out/t27.log
timeout 300 python scratch/dbg_mlp.py 2>&1 | tail -5
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b27.json 2> gpurun_out/b27.err; echo "bench $?"
python -c "
import json; d=json.load(open('gpurun_out/b27.json')); print(d['value'], d['ms_per_step'], d['config']['tuned_gemm_table'], d['config']['peak_hbm_gib'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
tail -2 gpurun_out/b27.err
