#!/bin/bash
cd $GRAFT_REPO_ROOT
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --latency-batches= --host-fed-steps 0 --sub-records="
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; }
for lib in "" scratch/r04/lib_heavy_first.so scratch/r04/lib_light_first.so "" scratch/r04/lib_heavy_first.so scratch/r04/lib_light_first.so; do
echo "LIB=$lib"
VER_HIP_LIB=${lib:+$PWD/$lib} timeout 600 python bench.py $ARGS 2>/dev/null | show
done
VER_HIP_LIB=$PWD/scratch/r04/lib_heavy_first.so timeout 600 python -m pytest tests/test_hip_ops_gpu.py -x -q -k "sca_gather_bf16 or launch_modes" 2>&1 | tail -2
