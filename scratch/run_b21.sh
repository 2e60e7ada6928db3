cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b21.json 2> gpurun_out/b21.err; echo "bench $?"
python -c "
import json; d=json.load(open('gpurun_out/b21.json')); print(d['value'], d['ms_per_step'], d['config']['tuned_gemm_table'], d['roofline']['frac'])"
tail -2 gpurun_out/b21.err
