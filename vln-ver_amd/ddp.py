"""Data-parallel training of the lifting path: one process per GPU, viewpoints sharded across
ranks, gradients summed with one bucketed all-reduce (RCCL over xGMI through
``torch.distributed``; ``gloo`` on CPU for the logic tests).

The path shards by independent units (a viewpoint = 6 views -> one volume); there is no data-path
collective, only the gradient sum.  Reference behaviour being mirrored:
* ``MMDistributedDataParallel(model, broadcast_buffers=False, find_unused_parameters=True)``
  (apis/mmdet_train.py:71-80) -> torch DDP; we freeze what the step does not touch instead of
  paying ``find_unused_parameters``;
* ``DistributedGroupSampler`` (datasets/samplers/group_sampler.py:62-103): every rank derives the
  same seed+epoch permutation and takes its contiguous slice -> :func:`shard_indices`;
* ``reduce_mean`` of the loss normalisers (mmdet, head:954,964) -> :func:`reduce_mean`.
"""
import math

import torch
import torch.distributed as dist


def shard_indices(group_flags, world_size, rank, samples_per_gpu=1, seed=0, epoch=0):
    """Indices this rank visits in ``epoch`` -- same rule as the reference sampler: per group a
    ``torch.randperm`` (generator seeded with epoch+seed), padded by repetition to a multiple of
    samples_per_gpu*world_size, then a permutation of the samples_per_gpu-sized chunks, then the
    rank's contiguous slice."""
    flags = torch.as_tensor(group_flags, dtype=torch.long)
    sizes = torch.bincount(flags)
    g = torch.Generator()
    g.manual_seed(epoch + seed)
    num_samples = sum(int(math.ceil(int(s) / samples_per_gpu / world_size)) * samples_per_gpu for s in sizes)
    indices = []
    for i, size in enumerate(sizes.tolist()):
        if size == 0:
            continue
        where = torch.nonzero(flags == i).squeeze(1)
        ind = where[torch.randperm(size, generator=g)].tolist()
        extra = int(math.ceil(size / samples_per_gpu / world_size)) * samples_per_gpu * world_size - len(ind)
        tmp = list(ind)
        for _ in range(extra // size):
            ind.extend(tmp)
        ind.extend(tmp[:extra % size])
        indices.extend(ind)
    assert len(indices) == num_samples * world_size
    chunks = torch.randperm(len(indices) // samples_per_gpu, generator=g).tolist()
    indices = [indices[j] for c in chunks for j in range(c * samples_per_gpu, (c + 1) * samples_per_gpu)]
    off = num_samples * rank
    return indices[off:off + num_samples]


def reduce_mean(t):
    """mmdet.core.reduce_mean: all-reduce(sum)/world (identity when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.clone()
    dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t


def wrap_ddp(module, device=None, bucket_cap_mb=200, bf16_gradients=True):
    """DDP wrapper used by bench.py.  bf16-compressed buckets halve the bytes on the xGMI links
    (ring all-reduce time is set by one 153 GB/s link: 2*(N-1)/N * bytes / link_bw); 200 MB
    buckets keep each of the three 88 MB (bf16) up_sample gradients in its own reduce so the
    first ones overlap the rest of the backward."""
    ids = [device.index] if (device is not None and device.type == 'cuda') else None
    ddp = torch.nn.parallel.DistributedDataParallel(module, device_ids=ids, gradient_as_bucket_view=True,
                                                    bucket_cap_mb=bucket_cap_mb, broadcast_buffers=False)
    if bf16_gradients and ids is not None:
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
    return ddp
