cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_detector_gpu.py -x -q -m gpu 2>&1 | tail -3
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do timeout 600 python bench.py --workload vocc_full_train --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --latency-batches= --host-fed-steps 0 2>/dev/null | show; done
