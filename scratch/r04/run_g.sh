#!/bin/bash
BENCH_ARGS="--sub-records= --latency-batches= --host-fed-steps 0" bash scratch/run_tune_gemms.sh > gpurun_out/r04_tune_run3.txt 2>&1; tail -3 gpurun_out/r04_tune_run3.txt | cut -c1-200
cp gpurun_out/tunableop_next0.csv vln-ver_amd/tuning/tunableop_gfx950_vocc.csv
python -m pytest tests/test_head_gpu.py -m gpu -x -q > gpurun_out/r04_pytest6.txt 2>&1; tail -2 gpurun_out/r04_pytest6.txt
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-fed-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['latency'], d['config']['full_train'], d['config']['fp32'])"
