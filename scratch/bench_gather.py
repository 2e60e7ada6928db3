"""Micro-benchmark of the fused gather kernels (fwd/bwd) at the vocc shape."""
import sys, importlib, warnings, json
warnings.filterwarnings('ignore')
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import os
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
if os.environ.get('VER_LIB'): hip.LIB_PATH = os.environ['VER_LIB']
import cases
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
grid = (4, 15, 15) if len(sys.argv) < 3 else tuple(int(v) for v in sys.argv[2].split('x'))
iters = 20
BF16 = len(sys.argv) > 3 and sys.argv[3] == 'bf16'
dev = 'cuda'
z, h, w = grid; nq = z*h*w
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, z, h, w)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, 6, 196, 8, 96, device=dev, generator=g)
if BF16: value = value.to(torch.bfloat16)
offs = torch.randn(B, nq, 8, 8, 2, device=dev, generator=g) * 3
logits = torch.randn(B, nq, 8, 8, device=dev, generator=g)
if os.environ.get('VER_BENCH_RING'):      # the reference's initial offsets (spatial_cross_attention.py:255-270), uniform attention
    import math
    th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
    gdir = torch.stack([th.cos(), th.sin()], -1)
    gdir = gdir / gdir.abs().max(-1, keepdim=True)[0]
    ring = gdir[:, None, :] * torch.arange(1, 9, dtype=torch.float32)[None, :, None]       # [heads, points, 2]
    offs = ring.to(dev)[None, None].expand(B, nq, 8, 8, 2).contiguous()
    logits = torch.zeros(B, nq, 8, 8, device=dev)
gs = torch.randn(B, nq, 768, device=dev, generator=g)
sn = int(hit.vis_cnt.sum())
fwd_b = B*6*196*768*4 + sn*(128+64+768)*4
bwd_b = 2*B*6*196*768*4 + sn*5376
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
v = value.clone().requires_grad_(True); o = offs.clone().requires_grad_(True); l = logits.clone().requires_grad_(True)
HM = bool(os.environ.get('VER_BENCH_HM'))          # head-major value layout (contiguous tiles)
PZ = bool(os.environ.get('VER_BENCH_PREZERO'))     # zero fill on the side stream, outside the timed launches
if HM:
    value = value.permute(3, 0, 1, 2, 4).contiguous(); v = value.clone().requires_grad_(True)
_prep = hip.sca_prepare_slots(hit, 768) if PZ else None      # once: a fresh 350-MB buffer per call measures the allocator
torch.cuda.synchronize()
def fwd():
    return hip.sca_gather(value, offs, logits, hit, 14, 14, _prep, HM)
t_f = timeit(fwd)
s = hip.sca_gather(v, o, l, hit, 14, 14, None, HM)
t_b = timeit(lambda: torch.autograd.grad(s, [v, o, l], gs, retain_graph=True))
t_p = timeit(lambda: hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, z, h, w))
print(json.dumps(dict(bf16=BF16, B=B, grid=grid, sigma_n=sn, fwd_us=round(t_f,1), fwd_GBs=round(fwd_b/t_f/1e3,1), fwd_frac=round(fwd_b/t_f/1e3/8000,4),
                      bwd_us=round(t_b,1), bwd_GBs=round(bwd_b/t_b/1e3,1), bwd_frac=round(bwd_b/t_b/1e3/8000,4), project_us=round(t_p,1))))
