"""Helper of tests/test_z_bench_launch.py::test_two_ranks_allreduce_the_gradients_of_the_real_step.

    python ddp_grad_helper.py single <data_rank> <out.pt>        one process, the data of `data_rank`, gradients saved
    torchrun --nproc-per-node 2 ddp_grad_helper.py ddp <out.pt>  two gloo ranks on the one GPU, rank 0 saves ITS gradients
    torchrun --nproc-per-node 1 ddp_grad_helper.py rccl <out.pt> ONE rank on an RCCL ('nccl') communicator with the bf16
                                                                 compression hook bench.py uses for N > 1
    torchrun --nproc-per-node 2 ddp_grad_helper.py ddp_update <out>   the same two gloo ranks, followed by bench.py's real
                                                                 update -- optim.ClipAdamW on the DDP bucket views; every rank
                                                                 saves <out>.<rank>: digests + samples of its parameters, the
                                                                 clip norm it got back, its (all-reduced) gradients' norm
    python ddp_grad_helper.py single_update <a.pt> <b.pt> <out>  one process: the MEAN of the gradients saved by two `single`
                                                                 runs as .grad, one ClipAdamW step, the same record

Both run ONE step of bench.py's LiftTrainer (the real HIP path: custom autograd Functions, frozen parameters,
gradient_as_bucket_view) in fp32 with dropout p = 0 at two viewpoints per rank, from the same seeded weights."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    mode = sys.argv[1]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    rank, world = 0, 1
    if mode == 'single_update':
        return single_update(dev, *sys.argv[2:5])
    if mode in ('ddp', 'rccl', 'ddp_update'):
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if mode == 'rccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
        rank, world = dist.get_rank(), dist.get_world_size()
        data_rank, out = rank, sys.argv[2]
    else:
        data_rank, out = int(sys.argv[2]), sys.argv[3]
    args = argparse.Namespace(config=None, workload='vocc_c2f_train', dtype='fp32')
    pkg, syn, head, _ = bench.build_model(args, dev)           # torch.manual_seed(2) + init_weights: same weights everywhere
    for m in head.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model = bench.LiftTrainer(head, 2, 'fp32').to(dev).train()
    net = model
    if mode != 'single':        # (bench.py: bf16_gradients = backend == 'nccl')
        net = importlib.import_module('vln-ver_amd.ddp').wrap_ddp(model, device=dev, bf16_gradients=mode == 'rccl')
    B = 2
    w2p_np, org_np = syn.camera_batch(B, seed=1 + data_rank)
    feats = torch.from_numpy(syn.vit_features(B, seed=100 + data_rank)).to(dev).permute(1, 0, 2, 3).contiguous()
    w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
    gt = torch.from_numpy(np.random.default_rng(7 + data_rank).integers(0, 17, size=(B, head.voxel_num))).to(dev)
    loss = net(feats, w2p, org, gt)
    loss.backward()
    torch.cuda.synchronize()
    if mode == 'ddp_update':
        torch.save(update_record(model), '%s.%d' % (out, rank))
    elif rank == 0:
        torch.save(dict(loss=float(loss), grads={k: p.grad.detach().float().cpu() for k, p in model.named_parameters()
                                                 if p.requires_grad and p.grad is not None}), out)
    if mode != 'single':
        dist.barrier()
        dist.destroy_process_group()


MAX_NORM = 1e-3         # (well under the step's gradient norm: the clip factor is active)


def update_record(model):
    """One ``optim.ClipAdamW`` step (bench.py's make_optimizer: lr 1e-4, weight decay 0.01) over the parameters that hold a
    gradient; -> sha256 of every updated parameter's bytes, a strided sample of it, the returned clip norm and the norm of the
    gradients the step read."""
    import hashlib
    params = [(k, p) for k, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    grad_norm = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for _, p in params)))
    views = sum(int(p.grad._is_view()) for _, p in params)
    opt = importlib.import_module('vln-ver_amd.optim').ClipAdamW([p for _, p in params], lr=1e-4, weight_decay=0.01, max_norm=MAX_NORM)
    norm = float(opt.step())
    torch.cuda.synchronize()
    rec = dict(clip_norm=norm, grad_norm=grad_norm, grad_views=views, digest={}, sample={})
    for k, p in params:
        host = p.detach().cpu().contiguous()
        rec['digest'][k] = hashlib.sha256(host.numpy().tobytes()).hexdigest()
        rec['sample'][k] = host.reshape(-1)[::97].clone()
    return rec


def single_update(dev, a, b, out):
    args = argparse.Namespace(config=None, workload='vocc_c2f_train', dtype='fp32')
    pkg, syn, head, _ = bench.build_model(args, dev)
    model = bench.LiftTrainer(head, 2, 'fp32').to(dev).train()
    ga, gb = torch.load(a)['grads'], torch.load(b)['grads']
    for k, p in model.named_parameters():
        if k in ga:
            p.grad = (0.5 * (ga[k] + gb[k])).to(dev)
    torch.save(update_record(model), out)


if __name__ == '__main__':
    main()
