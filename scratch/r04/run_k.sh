#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_ops_gpu.py -x -q -k "sca_gather or launch_modes or sca_backward" 2>&1 | tail -3
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; }
for lib in "" scratch/r04/lib_nocount.so "" scratch/r04/lib_nocount.so ""; do
echo "LIB=$lib"
VER_HIP_LIB=${lib:+$PWD/$lib} timeout 600 python bench.py $ARGS 2>/dev/null | show
done
