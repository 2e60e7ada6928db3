"""Helper of tests/test_hip_ops_gpu.py::test_sca_gather_launch_modes: the launch shape of ver_sca_forward is read
from the environment once per process, so every mode runs in its own interpreter and writes its result to a file."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import cases  # noqa: E402


def case(seed, B, grid, heads, hd, ring):
    syn = importlib.import_module('vln-ver_amd.synthetic')
    hip = importlib.import_module('vln-ver_amd.hipops')
    rng = np.random.default_rng(seed)
    z, h, w = grid
    nq = z * h * w
    w2p, org = syn.camera_batch(B, seed=1)
    hit = hip.project_points(torch.from_numpy(w2p).cuda(), torch.from_numpy(org).cuda(), cases.PC_RANGE, z, h, w)
    value = torch.from_numpy(rng.standard_normal((B, 6, 196, heads, hd)).astype(np.float32)).cuda()
    logits = torch.from_numpy(rng.standard_normal((B, nq, heads, 8)).astype(np.float32)).cuda()
    if ring:        # the reference's initial offsets (spatial_cross_attention.py:255-270): many samples miss the map
        th = torch.arange(heads, dtype=torch.float32) * (2.0 * np.pi / heads)
        d = torch.stack([th.cos(), th.sin()], -1)
        d = d / d.abs().max(-1, keepdim=True)[0]
        ring_o = d[:, None, :] * torch.arange(1, 9, dtype=torch.float32)[None, :, None]
        offsets = ring_o.cuda()[None, None].expand(B, nq, heads, 8, 2).contiguous()
    else:
        offsets = torch.from_numpy((rng.standard_normal((B, nq, heads, 8, 2)) * 3.0).astype(np.float32)).cuda()
    return hip, hit, value, offsets, logits


if __name__ == '__main__':
    out = sys.argv[1]
    res = {}
    for name, args in (('vocc_f32', (5, 3, (4, 15, 15), 8, 96, False)), ('vocc_ring', (6, 3, (4, 15, 15), 8, 96, True)),
                       ('small', (7, 2, (3, 7, 6), 2, 32, False))):
        hip, hit, value, offsets, logits = case(*args)
        res[name + '_f32'] = hip.sca_gather(value, offsets, logits, hit, 14, 14).cpu().numpy()
        res[name + '_bf16'] = hip.sca_gather(value.bfloat16(), offsets, logits, hit, 14, 14).cpu().numpy()
    np.savez(out, **res)
