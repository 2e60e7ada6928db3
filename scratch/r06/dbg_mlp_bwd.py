"""Which outputs of the wave-specialised backward differ between the saved-statistics (4-row mapping) form and the recomputing form."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
hip = importlib.import_module('vln-ver_amd.hipops')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = 'cuda'
gen = torch.Generator(device='cpu').manual_seed(300)
def P(*s, sc=0.1): return torch.randn(*s, generator=gen) * sc
p = dict(w1=P(128, 128), b1=P(128), g1=1 + P(128), be1=P(128), w2=P(128, 128), b2=P(128), g2=1 + P(128), be2=P(128), w3=P(16, 128), b3=P(16))
keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
x0 = torch.randn(n, 128, generator=gen) * 1.5
gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
def center(w, b): return w - w.mean(0, keepdim=True), b - b.mean()
res = {}
for saved in (True, False):
    hip._OCC_MLP_SAVE_RSTD = saved
    pd = {k: p[k].to(DEV).requires_grad_(True) for k in ('w1', 'b1') + keys}
    w1, b1 = center(pd['w1'], pd['b1']); w2, b2 = center(pd['w2'], pd['b2'])
    a1 = (x0.to(DEV) @ w1.t() + b1).bfloat16().detach().requires_grad_(True)
    out = hip.occ_mlp(a1, None, None, pd['g1'], pd['be1'], w2, b2, pd['g2'], pd['be2'], pd['w3'], pd['b3'], centered=True)
    out.backward(gy.to(DEV))
    res[saved] = (out.detach().float().cpu(), a1.grad.float().cpu(), {k: pd[k].grad.float().cpu() for k in keys})
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
print('dx', rel(res[True][1], res[False][1]))
for k in keys: print(k, rel(res[True][2][k], res[False][2][k]))
a, b = res[True][1], res[False][1]
print('per row (first 8):', [round(rel(a[r], b[r]), 3) for r in range(min(8, n))])
print('per 8-feature chunk:', [round(rel(a[:, 8 * c:8 * c + 8], b[:, 8 * c:8 * c + 8]), 3) for c in range(16)])
print('ratio sample', (a[0, :8] / b[0, :8]).tolist())
