"""Timeline of the loader-wave form of k_sca_fwd_cs (VER_SCA_CS_NLOAD > 0): per tile, when the loader wave has issued /
landed / converted / passed the barrier, and when the consumer waves reach / pass the tile barrier."""
import sys, importlib, ctypes, os, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, numpy as np
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
import cases
hip.LIB_PATH = os.path.abspath(os.environ['VER_LIB'])
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
nl = int(os.environ.get('VER_SCA_CS_NLOAD', '0'))
for pr in range(0, NP, 5):
    t0 = t[pr, :nw, 0][t[pr, :nw, 0] > 0].min() if (t[pr, :nw, 0] > 0).any() else 0
    if t0 == 0: continue
    print('== probe', pr)
    for i in range(5):
        lw = nw - 1
        L = [int(t[pr, lw, 8 * i + k] - t0) for k in (10, 11, 12, 13)]
        cb = [int(t[pr, w, 8 * i + 14] - t0) for w in range(nw - nl)]
        ca = [int(t[pr, w, 8 * i + 15] - t0) for w in range(nw - nl)]
        print(' tile %d loader: top %d landed %d converted %d barrier %d | consumers reach barrier min %d max %d, pass %d' % (i, L[0], L[1], L[2], L[3], min(cb), max(cb), max(ca)))
