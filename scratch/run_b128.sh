cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "192 64" "256 64"; do set -- $cfg
timeout 600 python bench.py --batch $1 --micro $2 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/b_$1_$2.json 2> gpurun_out/b_$1_$2.err; echo "bench $1/$2 rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/b_$1_$2.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('peak_hbm_gib'))"
done
