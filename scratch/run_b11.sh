cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python scratch/torch_prof.py --batch 8 --micro 8 > gpurun_out/tp.log 2>&1; echo "prof $?"
timeout 400 python scratch/exp_layer0.py > gpurun_out/exp_layer0.log 2>&1; echo "exp $?"
timeout 300 python bench.py --steps 3 --warmup 1 --batch 32 --micro 16 --no-cpu-baseline > gpurun_out/b_m16.json 2> gpurun_out/b_m16.err; echo "m16 $?"
timeout 300 python bench.py --steps 3 --warmup 1 --batch 32 --micro 32 --no-cpu-baseline > gpurun_out/b_m32.json 2> gpurun_out/b_m32.err; echo "m32 $?"
tail -3 gpurun_out/tp.log; cat gpurun_out/exp_layer0.log; cut -c1-200 gpurun_out/b_m16.json gpurun_out/b_m32.json; tail -2 gpurun_out/b_m32.err
