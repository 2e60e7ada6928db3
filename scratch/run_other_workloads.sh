cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --workload vocc_full_train --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b22_full32.json 2> gpurun_out/b22_full32.err; echo "full32 $?"
python -c "
import json; d=json.load(open('gpurun_out/b22_full32.json')); print(d['value'], d['ms_per_step'])"
timeout 600 python bench.py --workload vocc_full_train --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b22_full64.json 2> gpurun_out/b22_full64.err; echo "full64 $?"
python -c "
import json; d=json.load(open('gpurun_out/b22_full64.json')); print(d['value'], d['ms_per_step'])"
timeout 600 python bench.py --workload c2_single_scale_fwd --dtype fp32 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b22_c2.json 2> gpurun_out/b22_c2.err; echo "c2 $?"
python -c "
import json; d=json.load(open('gpurun_out/b22_c2.json')); print(d['value'], d['ms_per_step'])"
timeout 600 python bench.py --dtype fp32 --batch 8 --micro 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b22_fp32.json 2> gpurun_out/b22_fp32.err; echo "fp32 $?"
python -c "
import json; d=json.load(open('gpurun_out/b22_fp32.json')); print(d['value'], d['ms_per_step'])"
