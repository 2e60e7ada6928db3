"""Fused occupancy-MLP kernels alone on N rows (default 96 768 000 = the 192-viewpoint step): forward (+ saved statistics) and the
wave-specialised backward, HIP events.  VER_OCC_MLP_ROWS4 / VER_OCC_MLP_SAVE_RSTD select the backward's form (read at start).
    python scratch/r05/occ_mlp_time.py [N]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96768000
dev = 'cuda'
g = torch.Generator(device='cpu').manual_seed(0)
def P(*s): return (torch.randn(*s, generator=g) * 0.1).to(dev).requires_grad_(True)
w2, b2, w3, b3 = P(128, 128), P(128), P(16, 128), P(16)
g1, be1, g2, be2 = (1 + P(128)).detach().requires_grad_(True), P(128), (1 + P(128)).detach().requires_grad_(True), P(128)
w2c = (w2 - w2.mean(0, keepdim=True)); b2c = b2 - b2.mean()
x = torch.randn(N // 8, 128, device=dev).repeat(8, 1)
x = (x - x.mean(1, keepdim=True)).to(torch.bfloat16).requires_grad_(True)
gy = (torch.randn(N, 16, device=dev) * 0.1).to(torch.bfloat16)
def step():
    out = hip.occ_mlp(x, None, None, g1, be1, w2c, b2c, g2, be2, w3, b3, centered=True)
    out.backward(gy)
    x.grad = None
for _ in range(2): step()
torch.cuda.synchronize()
timer = hip.KernelTimer(); hip.KERNEL_TIMER = timer
for _ in range(4): step()
hip.KERNEL_TIMER = None
kt = timer.summary()
print('rows %d ROWS4=%s SAVE_RSTD=%s:' % (N, os.environ.get('VER_OCC_MLP_ROWS4', '1'), os.environ.get('VER_OCC_MLP_SAVE_RSTD', '1')),
      ', '.join('%s %.2f ms' % (k, v['ms'] / v['count']) for k, v in kt.items() if 'occ_mlp' in k))
