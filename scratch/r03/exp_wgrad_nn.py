"""Weight-gradient GEMMs a^T g as the library's TN form (current rows_tn) against: transpose the narrow operand first, then NN."""
import sys, torch
dev = 'cuda'
def t(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
shapes = [(552960, 824, 4480), (552960, 728, 4480), (345600, 14304, 1536), (345600, 9536, 1536), (345600, 6304, 1536), (86400, 38400, 1536)]
for (m, k, n) in shapes:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    g = torch.randn(m, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * k * n
    def tn(s=8):
        a3 = a.unflatten(0, (s, m // s)); g3 = g.unflatten(0, (s, m // s))
        return torch.bmm(a3.transpose(1, 2), g3).sum(0, dtype=torch.float32).to(a.dtype)
    ref = tn().float()
    line = 'M=%d K=%d N=%d: TN s=8 %.2f ms' % (m, k, n, t(tn))
    # transpose the narrower operand; output comes out transposed or not accordingly
    narrow_a = k <= n
    src = a if narrow_a else g
    tt = t(lambda: src.t().contiguous())
    line += ' | transpose %s %.2f ms' % ('a' if narrow_a else 'g', tt)
    st = src.t().contiguous()                      # [k or n, m]
    other = g if narrow_a else a
    for s in (1, 4, 8, 16):
        if m % s: continue
        def nn():
            if s == 1:
                return torch.mm(st, other)
            st3 = st.unflatten(1, (s, m // s)).permute(1, 0, 2)     # [s, rows, m/s]  (strided view)
            o3 = other.unflatten(0, (s, m // s))                     # [s, m/s, cols]
            return torch.bmm(st3, o3).sum(0, dtype=torch.float32).to(a.dtype)
        out = nn().float()
        if not narrow_a: out = out.t()
        err = float((out - ref).abs().max() / ref.abs().max())
        ms = t(nn)
        line += ' | NN s=%d %.2f ms (%.2f PF, err %.0e)' % (s, ms, fl / ms / 1e12, err)
    print(line, flush=True)
    del a, g, st
