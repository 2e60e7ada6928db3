"""Time the fused occupancy-MLP kernels alone (N rows, folded first Linear)."""
import sys, importlib, os, torch
sys.path.insert(0, '.')
hip = importlib.import_module('vln-ver_amd.hipops')
if os.environ.get('VER_LIB'): hip.LIB_PATH = os.environ['VER_LIB']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 504000 * 64
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(n, 128, device='cuda', generator=g).bfloat16().requires_grad_(True)
W = lambda *s: (torch.randn(*s, device='cuda', generator=g) * 0.1).requires_grad_(True)
p = [None, None, W(128), W(128), W(128, 128), W(128), W(128), W(128), W(16, 128), W(16)]
if os.environ.get('UNFOLDED'): p[0], p[1] = W(128, 128), W(128)
gy = (torch.randn(n, 16, device='cuda', generator=g) * 0.1).bfloat16()
def run():
    y = hip.occ_mlp(x, *p); y.backward(gy)
for _ in range(2): run()
torch.cuda.synchronize()
hip.TIMER = hip.KernelTimer() if hasattr(hip, 'KernelTimer') else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
print('rows %d  fwd+bwd (incl. weight-gradient GEMMs) %.2f ms' % (n, e0.elapsed_time(e1) / 3))
