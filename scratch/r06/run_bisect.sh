cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
run() { tag=$1; shift; python bench.py --steps 2 --warmup 1 --sub-records "" --cpu-seconds 4 "$@" > $R/bisect_$tag.json 2> $R/bisect_$tag.err; echo "$tag rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r06/bisect_$tag.json').read().strip().splitlines()[-1]); cb=d['cpu_baseline']; print('   ', d['ms_per_step'], {k:cb[k] for k in ('loss_gpu','loss_oracle','rel_diff')})
except Exception as e: print('   no line', e)
PY
}
run nolat --latency-batches "" --host-fed-steps 0
run lat1 --latency-batches "1" --host-fed-steps 0
run lat18_eager --latency-batches "1,8" --host-fed-steps 0 --graph-max-batch 0
run hostfed --latency-batches "" --host-fed-steps 2
