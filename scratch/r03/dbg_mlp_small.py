import sys, os, importlib, torch
sys.path[:0] = ['.', 'tests', 'tests/golden']
from test_hip_ops_gpu import _occ_mlp_params
hip = importlib.import_module('vln-ver_amd.hipops')
F = torch.nn.functional
for n in (1, 63, 64, 65, 128, 191, 16385, 32839):
    gen = torch.Generator(device='cpu').manual_seed(100 + n % 97)
    p = _occ_mlp_params(gen)
    a1 = (torch.randn(n, 128, generator=gen) * 1.5).bfloat16()
    gy = (torch.randn(n, 16, generator=gen) * 0.1).bfloat16()
    keys = ('g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'w3', 'b3')
    pr = {k: (v.bfloat16().double() if k.startswith('w') else v.double()).requires_grad_(True) for k, v in p.items()}
    xr = a1.double().requires_grad_(True)
    h = F.relu(F.layer_norm(xr, (128,), pr['g1'], pr['be1'], 1e-5))
    h = F.relu(F.layer_norm(h @ pr['w2'].t() + pr['b2'], (128,), pr['g2'], pr['be2'], 1e-5))
    ((h @ pr['w3'].t() + pr['b3']) * gy.double()).sum().backward()
    out = []
    for fused in (True, False):
        hip._OCC_MLP_BWD_FUSED = fused
        pd = {k: p[k].cuda().requires_grad_(True) for k in keys}
        xd = a1.cuda().requires_grad_(True)
        hip.occ_mlp(xd, None, None, *(pd[k] for k in keys)).backward(gy.cuda())
        rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
        out.append('x %.3f ' % rel(xd.grad.float(), xr.grad) + ' '.join('%s %.3f' % (k, rel(pd[k].grad, pr[k].grad)) for k in keys))
    print(n, 'fused  ', out[0]); print(n, 'rowsplt', out[1])
