cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-tuned-gemms > gpurun_out/b30.json 2> gpurun_out/b30.err; echo "bench $?"
python -c "
import json; d=json.load(open('gpurun_out/b30.json')); print(d['value'], d['ms_per_step'], d['config']['tuned_gemm_table'], d['config']['peak_hbm_gib'])"
timeout 400 python bench.py --steps 3 --warmup 0 --no-cpu-baseline > gpurun_out/b30b.json 2> gpurun_out/b30b.err; echo "bench $?"
python -c "
import json; d=json.load(open('gpurun_out/b30b.json')); print(d['value'], d['ms_per_step'], d['config']['tuned_gemm_table'], d['config']['peak_hbm_gib'])"
