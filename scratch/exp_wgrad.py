import torch
dev='cuda'
def t(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
M, Kt, N = 115200, 14464, 1536
A = torch.randn(M, Kt, device=dev, dtype=torch.bfloat16)
G = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
for (c0, c1) in ((0, 14304), (4928, 14464), (1696, 11232), (8160, 14464)):
    Av = A[:, c0:c1]
    fl = 2.0 * M * (c1 - c0) * N
    ref = torch.mm(Av.t(), G).float()
    base = t(lambda: torch.mm(Av.t(), G))
    line = 'K=%d mm %.2f ms %.0f TF |' % (c1 - c0, base, fl / base / 1e9)
    for S in (2, 4, 8):
        A3 = A.view(S, M // S, Kt)[:, :, c0:c1]          # strided batched view, no copy
        G3 = G.view(S, M // S, N)
        try:
            out = torch.bmm(A3.transpose(1, 2), G3).sum(0, dtype=torch.float32)
            err = float((out - ref).abs().max() / ref.abs().max())
            ms = t(lambda: torch.bmm(A3.transpose(1, 2), G3).sum(0, dtype=torch.float32))
            line += ' S=%d %.2f ms %.0f TF err %.1e |' % (S, ms, fl / ms / 1e9, err)
        except Exception as ex:
            line += ' S=%d FAILED %s |' % (S, type(ex).__name__)
    print(line)
