"""Phase timeline (s_memtime) of 32 probe workgroups of k_sca_bwd_mm (bf16 grad rows, as the bench runs it), third chunk of each;
library built with -DVER_DEBUG_TIMELINE into scratch/r06/lib_sca_timeline.so.
    python scratch/r06/timeline_bwd.py"""
import sys, importlib, ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests/golden'))
import torch, numpy as np
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
import cases
hip.LIB_PATH = os.path.abspath(os.environ.get('VER_LIB', 'scratch/r06/lib_sca_timeline.so'))
B = 192; dev = 'cuda'
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, 4, 15, 15)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, 6, 196, 8, 96, device=dev, generator=g).bfloat16().requires_grad_(True)
offs = (torch.randn(B, 900, 8, 8, 2, device=dev, generator=g) * 3).requires_grad_(True)
logits = torch.randn(B, 900, 8, 8, device=dev, generator=g).requires_grad_(True)
gs = torch.randn(B, 900, 768, device=dev, generator=g).bfloat16()
s = hip.sca_gather(value, offs, logits, hit, 14, 14, lowp_out=True) if 'lowp_out' in hip.sca_gather.__code__.co_varnames else hip.sca_gather(value, offs, logits, hit, 14, 14)
if s.dtype != torch.bfloat16: gs = gs.float()
timer = hip.KernelTimer(); hip.KERNEL_TIMER = timer
for _ in range(3): torch.autograd.grad(s, [value, offs, logits], gs, retain_graph=True)
hip.KERNEL_TIMER = None
torch.cuda.synchronize()
kt = timer.summary()
print({k: round(v['ms'] / v['count'] * 1e3, 1) for k, v in kt.items()}, 'us per launch; grad rows', gs.dtype)
NP, NW = 32, 16
N = NP * NW * 64
out = (ctypes.c_longlong * N)()
hip.lib().ver_timeline_read(out, N)
t = np.array(list(out), dtype=np.int64).reshape(NP, NW, 64)
names = ['start', 'chunk top', 'ops requested', 'B1 (G staged)', 'D done', 'B2', 'sample done', 'B3', 'zeroed + B4', 'scatter done', 'B5', 'dV done', 'B6', 'loop end']
rows = []
for pr in range(NP):
    if t[pr, 0, 0] == 0 or t[pr, 0, 12] == 0: continue
    for w in range(4):
        rows.append(t[pr, w, :14] - t[pr, :4, 0].min())
rows = np.array(rows, dtype=np.float64)
print('waves sampled', len(rows))
d = np.diff(rows[:, 1:13], axis=1)
for i in range(11):
    print('%-16s -> %-16s %7.0f ticks' % (names[1 + i], names[2 + i], d[:, i].mean()))
print('third chunk: %.0f ticks; workgroup lifetime %.0f ticks (%.1f chunks of that length)' % (d.sum(1).mean(), rows[:, 13].mean(), rows[:, 13].mean() / d.sum(1).mean()))
