cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python scratch/r05/wgrad_bench.py check sweep > gpurun_out/r05/wgrad2.txt 2>&1; echo "rc $?"
grep -c OK gpurun_out/r05/wgrad2.txt; grep "FAIL\|Error\|error" gpurun_out/r05/wgrad2.txt | head; grep "ver_wgrad_tn\|library" gpurun_out/r05/wgrad2.txt
timeout 1200 python -m pytest tests/test_head_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --sub-records= --latency-batches= --host-fed-steps 0 > gpurun_out/r05/bench2.json 2> gpurun_out/r05/bench2.err; echo "bench rc $?"; cut -c1-400 gpurun_out/r05/bench2.json; tail -3 gpurun_out/r05/bench2.err
