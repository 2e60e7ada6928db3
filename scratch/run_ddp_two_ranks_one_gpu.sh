cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 16 --micro 16 --backend gloo > gpurun_out/b29_ddp.json 2> gpurun_out/b29_ddp.err; echo "ddp $?"
cut -c1-400 gpurun_out/b29_ddp.json; tail -3 gpurun_out/b29_ddp.err
