# The N>1 path of bench.py with two ranks sharing the box's one GPU (gloo; the driver runs the real RCCL scaling).
# Both workloads: the lifting path (default) and the full multi-task head (DDP must not meet unused parameters).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for WL in vocc_c2f_train vocc_full_train; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 --micro 16 --backend gloo --workload $WL > gpurun_out/ddp2_$WL.json 2> gpurun_out/ddp2_$WL.err; echo "ddp $WL rc=$?"
  cut -c1-300 gpurun_out/ddp2_$WL.json; tail -3 gpurun_out/ddp2_$WL.err
done
