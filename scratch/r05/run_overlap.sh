cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/ovl; mkdir -p $R
timeout 900 python -m pytest tests/test_head_gpu.py -x -q > $R/tests.log 2>&1; echo "tests $?"; tail -12 $R/tests.log
FT="bench.py --workload vocc_full_train --steps 4 --warmup 1 --no-cpu-baseline --latency-batches= --host-fed-steps 0"
for m in 0 1; do
VER_LOWP_PARAMS=$m timeout 300 python $FT > $R/ft$m.json 2> $R/ft$m.err; echo "bench $?"
python -c "
import json;d=json.loads(open('$R/ft$m.json').read().strip().splitlines()[-1]);print('line', $m, d['value'], d['ms_per_step'])"
done
timeout 400 rocprofv3 --kernel-trace -d $R/t0 -o trace -- python3 $FT > $R/t0.json 2> $R/t0.err; echo "trace $?"
python scratch/r05/timeline.py $R/t0/trace_results.db 1 > $R/timeline_0.txt; rm -rf $R/t0
