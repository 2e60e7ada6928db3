"""Helpers shared by the test modules."""
import importlib
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def state_from(npz, prefix='sd.'):
    return {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}


def pkg(sub=None):
    name = 'vln-ver_amd' + ('.' + sub if sub else '')
    return importlib.import_module(name)


def oracle():
    return importlib.import_module('oracle.ver_oracle')


def maxdiff(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max())


def relerr(a, b, atol=1e-4):
    """max |a-b| / (atol + |b|*atol/1e-4*1e-5 ...) -- see ``close``."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float(((a - b).abs() / (1.0 + b.abs())).max())


def close(a, b, atol=1e-4, rtol=1e-5):
    """Gradient tolerance: |a-b| <= atol + rtol*|b| everywhere (values reach O(100), where
    fp32 summation-order noise alone is ~1e-5 relative)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    bad = (a - b).abs() > atol + rtol * b.abs()
    if bool(bad.any()):
        d = (a - b).abs()
        i = int(d.flatten().argmax())
        print('close(): %d of %d outside atol=%g rtol=%g; worst |a-b|=%.3e at flat %d (a=%.6e b=%.6e)'
              % (int(bad.sum()), bad.numel(), atol, rtol, float(d.flatten()[i]), i,
                 float(a.flatten()[i]), float(b.flatten()[i])))
        return False
    return True


def oracle_slots(o, value, offsets, logits, uv, mask, map_hw):
    """Camera-averaged gather ``slots`` [B,Nq,C] composed from the oracle's core op exactly as
    the reference composes it (re-batch visible voxels per camera -> MSDA -> scatter-add ->
    divide by the camera count; spatial_cross_attention.py:139-173,345-398).
    value [B,Ncam,Nk,heads,hd]; offsets [B,Nq,heads,P,2] (pixels); logits [B,Nq,heads,P];
    uv [B,Ncam,Nq,D,2]; mask bool [B,Ncam,Nq]."""
    B, ncam, nk, heads, hd = value.shape
    nq, P, D = offsets.shape[1], offsets.shape[3], uv.shape[3]
    mh, mw = map_hw
    norm = torch.tensor([float(mw), float(mh)], dtype=value.dtype)
    out = []
    for b in range(B):
        aw = logits[b].softmax(-1)
        off = (offsets[b] / norm).view(nq, heads, P // D, D, 2)
        slots = value.new_zeros(nq, heads * hd)
        for c in range(ncam):
            idx = mask[b, c].nonzero().squeeze(-1)
            if idx.numel() == 0:
                continue
            loc = (uv[b, c, idx][:, None, None, :, :] + off[idx]).reshape(len(idx), heads, 1, P, 2)
            got = o.msda_core(value[b, c][None], [(mh, mw)], loc[None], aw[idx][None, :, :, None, :])
            slots = slots.index_add(0, idx, got[0])
        count = mask[b].sum(0).clamp(min=1).to(value.dtype)
        out.append(slots / count[:, None])
    return torch.stack(out)


def close_mostly(a, b, atol=2e-4, rtol=1e-4, max_bad_frac=0.05, max_rel_l2=2e-2, max_median=2e-5):
    """Gradient check for whole-network backward passes.

    The network has kinks (ReLU pre-activations within ~1e-6 of zero, bilinear cell edges): with
    ~4e6 hidden activations a few of them flip between ANY two fp32 implementations (the CPU
    oracle on two different hosts differs from itself by 1.5e-3 in d(feats)), and one flipped
    unit changes one voxel's gradient by O(0.1), which then spreads over a camera's value tile.
    So: median error tiny, few elements outside the element-wise tolerance, small relative L2.
    Kernel arithmetic itself is checked element-wise in test_hip_ops_gpu.py (no kinks there)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    d = (a - b).abs()
    bad = float((d > atol + rtol * b.abs()).double().mean())
    rel = float(d.norm() / b.norm().clamp(min=1e-30))
    med = float(d.median())
    ok = bad <= max_bad_frac and rel <= max_rel_l2 and med <= max_median
    if not ok:
        print('close_mostly(): bad fraction %.4f, rel L2 %.3e, median %.3e' % (bad, rel, med))
    return ok


def rel_l2(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def init_report(head):
    """What ``init_weights`` did to a freshly built head, in a form two implementations can be compared by:
    per state-dict entry whether it was re-initialised (differs from its construction-time value), whether the
    result is a constant, its value if so, and its std / |max| otherwise."""
    before = {k: v.detach().clone() for k, v in head.state_dict().items()}
    head.init_weights()
    names, changed, const, value, std, amax = [], [], [], [], [], []
    for k, v in head.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        v = v.detach().double()
        names.append(k)
        changed.append(bool((v != before[k].double()).any()))
        c = bool((v == v.flatten()[0]).all())
        const.append(c)
        value.append(float(v.flatten()[0]) if c else float('nan'))
        std.append(float(v.std()) if v.numel() > 1 else 0.0)
        amax.append(float(v.abs().max()))
    return dict(names=np.array(names), changed=np.array(changed), const=np.array(const),
                value=np.array(value, dtype=np.float64), std=np.array(std, dtype=np.float64),
                amax=np.array(amax, dtype=np.float64))
