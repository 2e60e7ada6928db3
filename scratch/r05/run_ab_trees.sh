cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/abt; mkdir -p $R
for rep in 1 2; do
for t in ab_old .; do
( cd $GRAFT_REPO_ROOT/$t && timeout 600 python bench.py --no-cpu-baseline --sub-records=full_train --host-fed-steps 0 --latency-batches 1,8 --steps 6 --warmup 2 > $R/b.json 2> $R/b.err; python -c "
import json;d=json.loads(open('$R/b.json').read().strip().splitlines()[-1]);c=d['config'];print('tree [$t]', d['value'], d['ms_per_step'], [l['ms_per_step'] for l in c.get('latency')], c['full_train']['viewpoints_per_s'])" )
done
done
