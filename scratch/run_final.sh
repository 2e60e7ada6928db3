cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t_final.log 2>&1; echo "tests $?"; tail -3 gpurun_out/t_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/b_final.json 2> gpurun_out/b_final.err; echo "bench $?"
python -c "
import json; d=json.load(open('gpurun_out/b_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_other_kernels'][0]['frac'], d['cpu_baseline']['value'])"
