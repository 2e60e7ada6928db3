"""GEMM shape experiments for the upsample layers (scratch)."""
import torch, time
dev = torch.device('cuda')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
K, N = 57600, 768
for M in (7200, 14400, 28800):
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(K, N, device=dev, dtype=torch.bfloat16) * 0.01
    bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    ms = t(lambda: torch.addmm(bias, A, W)); print('M=%d full addmm %.3f ms %.0f TF' % (M, ms, fl / ms / 1e9))
    for S in (3, 5, 15, 25):
        Ks = K // S
        As = A.view(M, S, Ks).permute(1, 0, 2).contiguous(); Ws = W.view(S, Ks, N)
        ms = t(lambda: torch.bmm(As, Ws).sum(0)); print('   bmm S=%d %.3f ms %.0f TF' % (S, ms, fl / ms / 1e9))
        try:
            ms = t(lambda: torch.bmm(As, Ws, out_dtype=torch.float32).sum(0)); print('   bmm f32out S=%d %.3f ms %.0f TF' % (S, ms, fl / ms / 1e9))
        except Exception as ex:
            print('   out_dtype unsupported', type(ex).__name__)
        def seq():
            o = torch.addmm(bias, As[0], Ws[0])
            for s in range(1, S): o = torch.addmm(o, As[s], Ws[s])
            return o
        if S <= 5:
            ms = t(seq); print('   seq S=%d %.3f ms %.0f TF' % (S, ms, fl / ms / 1e9))
    G = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: G @ W.t()); print('   dgrad %.3f ms %.0f TF' % (ms, fl / ms / 1e9))
    ms = t(lambda: A.t() @ G); print('   wgrad %.3f ms %.0f TF' % (ms, fl / ms / 1e9))
    Wt = W.t().contiguous()
    ms = t(lambda: G @ Wt); print('   dgrad (W^T contiguous) %.3f ms %.0f TF' % (ms, fl / ms / 1e9))
    del A, W, G
# bias taps
k = torch.randn(75, 768, 768, device=dev, dtype=torch.bfloat16); pb = torch.randn(768, device=dev, dtype=torch.bfloat16)
print('bias taps matmul %.3f ms' % t(lambda: torch.matmul(pb, k)))
print('bias taps mul-sum %.3f ms' % t(lambda: (pb[None, :, None] * k).sum(1)))
print('bias taps einsum %.3f ms' % t(lambda: torch.einsum('i,tio->to', pb, k)))
print('bias taps f32 mv %.3f ms' % t(lambda: (k.permute(0, 2, 1).reshape(-1, 768) @ pb)))
# lattice-layer GEMMs
for M, Kt in ((8 * 3600, 27 * 768), (8 * 14400, 27 * 768), (8 * 14400, 12 * 768), (32 * 14400, 27 * 768)):
    A = torch.randn(M, Kt, device=dev, dtype=torch.bfloat16); W = torch.randn(Kt, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * Kt * N
    ms = t(lambda: A @ W); print('lattice M=%d K=%d %.3f ms %.0f TF' % (M, Kt, ms, fl / ms / 1e9))
    G = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: G @ W.t()); print('   dgrad %.3f ms %.0f TF' % (ms, fl / ms / 1e9))
    ms = t(lambda: A.t() @ G); print('   wgrad %.3f ms %.0f TF' % (ms, fl / ms / 1e9))
    del A, W, G
