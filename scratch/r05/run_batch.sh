cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/bt; mkdir -p $R
for b in 192 288 384; do
timeout 500 python bench.py --batch $b --micro $b --no-cpu-baseline --sub-records= --latency-batches= --host-fed-steps 0 --steps 3 --warmup 1 > $R/b$b.json 2> $R/b$b.err; echo "bench $b $?"; tail -2 $R/b$b.err | cut -c1-300
python -c "
import json;d=json.loads(open('$R/b$b.json').read().strip().splitlines()[-1]);print('line', $b, d['value'], d['ms_per_step'], d['config']['peak_hbm_gib'], d['roofline']['frac'], [ (o['kernel'], o['frac']) for o in d['roofline_other_kernels'] if o['kernel'] in ('head_gemms','whole_step')])"
done
