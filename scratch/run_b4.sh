timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/b4_b32.json 2> gpurun_out/b4.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --micro 4 --backend gloo --no-cpu-baseline > gpurun_out/b4_ddp_gloo.json 2> gpurun_out/b4_ddp.err
echo rc=$? >> gpurun_out/b4_ddp.err
