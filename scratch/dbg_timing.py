import sys, importlib, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import torch, numpy as np
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
import cases
hip.LIB_PATH = 'scratch/libver_hip_dbg.so'
B=64; dev='cuda'
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, 4,15,15)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B,6,196,8,96, device=dev, generator=g); offs = torch.randn(B,900,8,8,2, device=dev, generator=g)*3; logits = torch.randn(B,900,8,8, device=dev, generator=g)
for _ in range(3): hip.sca_gather(value, offs, logits, hit, 14, 14)
torch.cuda.synchronize()
out = (ctypes.c_longlong*256)()
hip.lib().ver_debug_read(out, 256)
t = list(out)
t0 = t[0]
print('zero_cnt', hit.zero_cnt.tolist(), 'vis_cnt', hit.vis_cnt[0].tolist())
print('start->loader issued first tile: %d ticks' % (t[1]-t0))
for hh in range(4):
    b = 8+hh*8
    print('head %d: loader tile landed @%d | w0 at barrier @%d | after barrier @%d | first operands issued @%d | main loop done @%d | slow loop done @%d' % tuple([hh]+[t[b+i]-t0 for i in range(6)]))
