import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_hip_ops_gpu as T
hip = importlib.import_module('vln-ver_amd.hipops')
DEV='cuda'
gen = torch.Generator(device='cpu').manual_seed(12)
p = T._occ_mlp_params(gen)
n = 8000 * 2 + 16 * 31 + 5
x = (torch.randn(n, 128, generator=gen) * 1.5).bfloat16()
gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
keys = ('w1', 'b1', 'g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
pr = {k: (v.bfloat16().double() if k.startswith('w') else v.double()).requires_grad_(True) for k, v in p.items()}
xr = x.double().requires_grad_(True)
ref = T._occ_mlp_reference(xr, pr, round_hidden=False); ref.backward(gy.double())
def rel(a, b): return float((a.double().cpu() - b).norm() / b.norm())
# fused
pd = {k: p[k].to(DEV).requires_grad_(True) for k in keys}
xd = x.to(DEV).requires_grad_(True)
out = hip.occ_mlp(xd, *(pd[k] for k in keys)); out.backward(gy.to(DEV))
print('fused  out %.4f dx %.4f' % (rel(out.detach().float(), ref.detach()), rel(xd.grad.float(), xr.grad)), {k: round(rel(pd[k].grad, pr[k].grad), 4) for k in keys})
# torch autocast layer by layer
F = torch.nn.functional
pt = {k: p[k].to(DEV).requires_grad_(True) for k in keys}
xt = x.to(DEV).requires_grad_(True)
with torch.autocast('cuda', dtype=torch.bfloat16):
    h = F.linear(xt, pt['w1'], pt['b1']); h = hip.layer_norm_relu(h, pt['g1'], pt['be1'])
    h = F.linear(h, pt['w2'], pt['b2']); h = hip.layer_norm_relu(h, pt['g2'], pt['be2'])
    o = F.linear(h, pt['w3'], pt['b3'])
o.backward(gy.to(DEV))
print('layers out %.4f dx %.4f' % (rel(o.detach().float(), ref.detach()), rel(xt.grad.float(), xr.grad)), {k: round(rel(pt[k].grad, pr[k].grad), 4) for k in keys})
# timing at scale
N = 504000 * 8
X = torch.randn(N, 128, device=DEV).bfloat16().requires_grad_(True)
G = (torch.randn(N, 16, device=DEV) * 0.1).bfloat16()
def run_fused():
    o = hip.occ_mlp(X, *(pd[k] for k in keys)); o.backward(G)
def run_layers():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        h = F.linear(X, pt['w1'], pt['b1']); h = hip.layer_norm_relu(h, pt['g1'], pt['be1'])
        h = F.linear(h, pt['w2'], pt['b2']); h = hip.layer_norm_relu(h, pt['g2'], pt['be2'])
        o = F.linear(h, pt['w3'], pt['b3'])
    o.backward(G)
for name, fn in (('fused', run_fused), ('layers', run_layers)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize(); print(name, 'fwd+bwd ms for 8 viewpoints: %.2f' % (s.elapsed_time(e) / 5))
t = hip.KernelTimer(); hip.KERNEL_TIMER = t
for _ in range(3): run_fused()
print({k: round(v['ms'] / v['count'], 3) for k, v in t.summary().items()})
