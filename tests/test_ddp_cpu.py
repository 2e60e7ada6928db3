"""N>1 logic on CPU with gloo, world_size 2: rank sharding rule, gradient averaging through the
DDP wrapper (on the CPU-runnable part of the path: the even-lattice occupancy head), reduce_mean."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import pkg


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _TinyOcc(torch.nn.Module):
    """The dense half of the path at toy width (pure torch, so it runs on CPU): even-lattice
    upsample of three ConvTranspose3d + a linear head."""

    def __init__(self, c=6):
        super().__init__()
        import importlib
        self.up = importlib.import_module('vln-ver_amd.dense_heads.upsample')
        geom = dict(stride=(1, 2, 2), padding=(2, 4, 4), dilation=(2, 2, 2), output_padding=(0, 1, 1))
        self.convs = torch.nn.ModuleList([torch.nn.ConvTranspose3d(c, c, (3, 5, 5), **geom) for _ in range(3)])
        self.head = torch.nn.Linear(c, 3)

    def forward(self, x):
        y = self.up.upsample_dense(x, [m.weight for m in self.convs], [m.bias for m in self.convs])
        return self.head(y.permute(0, 2, 3, 4, 1)).square().mean()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import importlib
        ddp_mod = importlib.import_module('vln-ver_amd.ddp')
        torch.manual_seed(0)
        model = _TinyOcc()
        ddp = ddp_mod.wrap_ddp(model, device=torch.device('cpu'))
        data = torch.from_numpy(np.random.default_rng(5).standard_normal((4, 6, 2, 3, 3)).astype(np.float32))
        idx = ddp_mod.shard_indices([0, 0, 0, 0], world, rank, samples_per_gpu=2, seed=3, epoch=1)
        loss = ddp(data[idx])
        loss.backward()
        grads = torch.cat([p.grad.flatten() for p in model.parameters()])
        rm = ddp_mod.reduce_mean(torch.tensor([float(rank + 1)]))
        torch.save(dict(idx=idx, grads=grads, rm=rm), os.path.join(out, 'r%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process_mean(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % k)) for k in range(world)]
    assert sorted(r[0]['idx'] + r[1]['idx']) == [0, 1, 2, 3]           # disjoint cover
    assert torch.allclose(r[0]['grads'], r[1]['grads'])                # same averaged gradient
    assert float(r[0]['rm']) == pytest.approx(1.5)
    torch.manual_seed(0)
    model = _TinyOcc()
    data = torch.from_numpy(np.random.default_rng(5).standard_normal((4, 6, 2, 3, 3)).astype(np.float32))
    total = 0.5 * (model(data[r[0]['idx']]) + model(data[r[1]['idx']]))
    total.backward()
    want = torch.cat([p.grad.flatten() for p in model.parameters()])
    assert torch.allclose(r[0]['grads'], want, atol=1e-6, rtol=1e-5)


def test_shard_rule_matches_reference_sampler_contract():
    ddp_mod = pkg('ddp')
    flags = [0] * 7 + [1] * 3
    world, spg = 4, 1
    per_rank = [ddp_mod.shard_indices(flags, world, rk, spg, seed=0, epoch=5) for rk in range(world)]
    n = len(per_rank[0])
    assert all(len(p) == n for p in per_rank)
    assert n == int(np.ceil(7 / world)) + int(np.ceil(3 / world))      # group_sampler.py:54-59
    flat = [i for p in per_rank for i in p]
    assert set(flat) == set(range(10))                                 # padded by repetition
    assert per_rank == [ddp_mod.shard_indices(flags, world, rk, spg, seed=0, epoch=5) for rk in range(world)]
    assert per_rank != [ddp_mod.shard_indices(flags, world, rk, spg, seed=0, epoch=6) for rk in range(world)]
