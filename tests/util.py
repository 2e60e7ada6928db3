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
    return bool(((a - b).abs() <= atol + rtol * b.abs()).all())
