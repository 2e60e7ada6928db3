import torch, time
dev='cuda'
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
M, C, Kt = 115200, 768, 27*768+320
G = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
W = torch.randn(6*C, C, device=dev, dtype=torch.bfloat16) * 0.05      # [K_range, Co]
dA = torch.zeros(M, Kt, device=dev, dtype=torch.bfloat16)
view = dA[:, 1000:1000+6*C]
ref = (G.float() @ W.float().t())
torch.mm(G, W.t(), out=view)
print('mm strided out ok:', torch.allclose(view.float(), ref, atol=0.5, rtol=2e-2), 'outside untouched:', float(dA[:, :1000].abs().max()), float(dA[:, 1000+6*C:].abs().max()))
torch.addmm(view, G, W.t(), out=view)
print('addmm accumulate ok:', torch.allclose(view.float(), 2*ref, atol=1.0, rtol=3e-2), 'outside:', float(dA[:, :1000].abs().max()))
print('mm strided-out ms', t(lambda: torch.mm(G, W.t(), out=view)))
tmp = torch.empty(M, 6*C, device=dev, dtype=torch.bfloat16)
print('mm contiguous-out ms', t(lambda: torch.mm(G, W.t(), out=tmp)))
print('addmm strided-out accumulate ms', t(lambda: torch.addmm(view, G, W.t(), out=view)))
# forward direction: A column-range view as input, accumulate into contiguous out
A = torch.randn(M, Kt, device=dev, dtype=torch.bfloat16)
Wf = torch.randn(6*C, C, device=dev, dtype=torch.bfloat16) * 0.05
out = torch.zeros(M, C, device=dev, dtype=torch.bfloat16)
Av = A[:, 1000:1000+6*C]
print('fwd strided-A mm ms', t(lambda: torch.mm(Av, Wf, out=out)), 'addmm', t(lambda: torch.addmm(out, Av, Wf, out=out)))
Ac = Av.contiguous()
print('fwd contiguous-A mm ms', t(lambda: torch.mm(Ac, Wf, out=out)))
# wgrad with strided A: A_range^T @ G
print('wgrad strided-A ms', t(lambda: torch.mm(Av.t(), G)), 'contig', t(lambda: torch.mm(Ac.t(), G)))
# full-K forward (27 blocks + aug)
Wfull = torch.randn(Kt, C, device=dev, dtype=torch.bfloat16) * 0.02
print('fwd full K=%d ms' % Kt, t(lambda: torch.mm(A, Wfull, out=out)))
Av18 = A[:, 9*C+80:]
W18 = torch.randn(Av18.shape[1], C, device=dev, dtype=torch.bfloat16) * 0.02
print('fwd K=%d (offset view) ms' % Av18.shape[1], t(lambda: torch.mm(Av18, W18, out=out)))
