"""s_memtime stamps of probe workgroups of k_sca_fwd_mm (library built with -DVER_DEBUG_TIMELINE)."""
import sys, importlib, ctypes, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, numpy as np
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
import cases
hip.LIB_PATH = os.path.abspath('scratch/r03/lib_timeline.so')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dev = 'cuda'
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, 4, 15, 15)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, 6, 196, 8, 96, device=dev, generator=g).bfloat16()
offs = torch.randn(B, 900, 8, 8, 2, device=dev, generator=g) * 3
logits = torch.randn(B, 900, 8, 8, device=dev, generator=g)
for _ in range(3):
    hip.sca_gather(value, offs, logits, hit, 14, 14)
torch.cuda.synchronize()
N = 4 * 16 * 64
out = (ctypes.c_longlong * N)()
hip.lib().ver_timeline_read(out, N)
t = np.array(list(out), dtype=np.int64).reshape(4, 16, 64)
names = ['start', 'prologue', 'done-head', 'bar-free', 'scat<', 'scat>', 'landed', 'converted', 'bar-tile', 'mfma>', 'stored', 'end']
for pr in range(2):
    t0 = (t[pr, :4, 0] >> 4).min()
    print('== probe workgroup %d (cycles since the first wave started; name +delta)' % pr)
    for w in range(int(os.environ.get('TL_WAVES', 2))):
        ev = [(int(x & 15), int(x >> 4) - t0) for x in t[pr, w] if x > 0]
        line, prev = [], 0
        for c, v in ev:
            line.append('%s +%d' % (names[c], v - prev)); prev = v
        print(' wave %d (end %d): ' % (w, prev) + ', '.join(line))
