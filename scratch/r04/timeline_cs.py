"""Phase timeline (s_memtime) of 32 probe workgroups of k_sca_fwd_cs (library built with -DVER_DEBUG_TIMELINE into
scratch/r04/lib_timeline.so): where a wave's cycles go between workgroup start and its last store."""
import sys, importlib, ctypes, os, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, numpy as np
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
import cases
hip.LIB_PATH = os.path.abspath(os.environ.get('VER_LIB', 'scratch/r04/lib_timeline.so'))
B = 192; dev = 'cuda'
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, 4, 15, 15)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, 6, 196, 8, 96, device=dev, generator=g).bfloat16()
th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
gdir = torch.stack([th.cos(), th.sin()], -1); gdir = gdir / gdir.abs().max(-1, keepdim=True)[0]
ring = gdir[:, None, :] * torch.arange(1, 9, dtype=torch.float32)[None, :, None]
offs = ring.to(dev)[None, None].expand(B, 900, 8, 8, 2).contiguous(); logits = torch.zeros(B, 900, 8, 8, device=dev)
for _ in range(3): s = hip.sca_gather(value, offs, logits, hit, 14, 14)
torch.cuda.synchronize()
NP, NW = 32, 16
N = NP * NW * 64
out = (ctypes.c_longlong * N)()
hip.lib().ver_timeline_read(out, N)
t = np.array(list(out), dtype=np.int64).reshape(NP, NW, 64)
nw = int(os.environ.get('VER_SCA_CS_THREADS_BF16', '256')) // 64
rows = []
for pr in range(NP):
    t0 = t[pr, :nw, 0].min()
    if t0 == 0: continue
    for w in range(nw):
        r = t[pr, w] - t0
        its = [r[8 + i] for i in range(10) if t[pr, w, 8 + i] > t[pr, w, 4]]
        rows.append(dict(start=r[0], issued=r[1], landed=r[2], conv=r[3], barrier=r[4], end=r[5], phaseA1=(r[6] - r[8]) if len(its) > 1 else -1,
                         pair0_1=(r[7] - r[6]) if len(its) > 1 else -1, iters=len(its), it_cycles=np.diff([r[4]] + its).tolist()))
def avg(k): 
    v = [x[k] for x in rows if x[k] >= 0]
    return sum(v) / max(1, len(v))
print('waves sampled', len(rows))
for k in ('start', 'issued', 'landed', 'conv', 'barrier', 'end', 'phaseA1', 'pair0_1', 'iters'):
    print('%-10s avg %9.0f' % (k, avg(k)))
full = [c for x in rows for c in x['it_cycles'][:-1]]
last = [x['it_cycles'][-1] for x in rows if x['it_cycles']]
print('full iteration avg cycles', sum(full) / max(1, len(full)), ' last iteration', sum(last) / max(1, len(last)))
ends = {}
for pr in range(NP):
    e = [(t[pr, w, 5] - t[pr, :nw, 0].min()) for w in range(nw) if t[pr, w, 0]]
    if e: ends[pr] = (min(e), max(e))
print('workgroup: first wave done / last wave done (avg)', np.mean([v[0] for v in ends.values()]), np.mean([v[1] for v in ends.values()]))
for x in rows[:12]: print(x)
