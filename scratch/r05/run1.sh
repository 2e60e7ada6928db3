cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python scratch/r05/wgrad_bench.py check big > gpurun_out/r05/wgrad1.txt 2>&1; echo "rc $?"
tail -40 gpurun_out/r05/wgrad1.txt
