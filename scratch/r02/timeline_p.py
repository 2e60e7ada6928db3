"""Per-tile barrier times of the persistent forward kernel (library built with -DVER_DEBUG_TIMELINE)."""
import sys, importlib, ctypes, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import torch, numpy as np
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
import cases
hip.LIB_PATH = os.path.abspath('scratch/r02/lib_timeline.so')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
dev='cuda'
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, 4,15,15)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B,6,196,8,96, device=dev, generator=g); offs = torch.randn(B,900,8,8,2, device=dev, generator=g)*3; logits = torch.randn(B,900,8,8, device=dev, generator=g)
if bf16: value = value.bfloat16()
for _ in range(3): hip.sca_gather(value, offs, logits, hit, 14, 14)
torch.cuda.synchronize()
N = 4*16*64
out = (ctypes.c_longlong*N)()
lib = hip.lib(); lib.ver_timeline_read(out, N)
t = np.array(list(out), dtype=np.int64).reshape(4, 16, 64)
for pr in range(2):
    t0 = t[pr, :, 0][t[pr, :, 0] > 0].min()
    print('== probe workgroup %d' % pr)
    for w in (0, 7, 14, 15):
        r = [int(x - t0) if x > 0 else None for x in t[pr, w]]
        tiles = [x for x in r[1:62] if x is not None]
        d = [tiles[i+1]-tiles[i] for i in range(len(tiles)-1)]
        print('wave %2d start %s; %s (tile %s): %s' % (w, r[0], 'landed' if w == 15 else 'after-barrier', 0, tiles[:3]), 'deltas', d)
