"""Layer-3 class GEMMs at the shipped K (multiples of 32) and at K padded to multiples of 64 (pattern blocks of 96 instead of 80
columns): forward [M, K] x [K, 1536], dgrad [M, 1536] x [1536, K] (into a column range of a wider matrix), wgrad 8-chunk bmm."""
import torch, json
import torch.cuda.tunable as tunable
dev = 'cuda'
tunable.enable(True); tunable.set_max_tuning_duration(20); tunable.tuning_enable(True)
tunable.set_filename('gpurun_out/r04_lattice_tunable.csv', True)
def timeit(fn, n=4):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, N = 345600, 1536
for K, LD in ((14304, 14464), (14400, 14592), (9376, 14464), (9408, 14592), (6304, 14464), (6336, 14592)):
    A = torch.randn(M, LD, device=dev, dtype=torch.bfloat16)
    a = A[:, :K]
    w = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dA = torch.empty(M, LD, device=dev, dtype=torch.bfloat16)
    fwd = timeit(lambda: torch.mm(a, w, out=out))
    dg = timeit(lambda: torch.mm(g, w.t(), out=dA[:, :K]))
    s = 8
    wg = timeit(lambda: torch.bmm(a.unflatten(0, (s, M // s)).transpose(1, 2), g.unflatten(0, (s, M // s))).sum(0, dtype=torch.float32))
    fl = 2.0 * M * N * K / 1e9
    print(json.dumps(dict(K=K, fwd_ms=round(fwd, 3), fwd_TF=round(fl / fwd), dgrad_ms=round(dg, 3), dgrad_TF=round(fl / dg), wgrad_ms=round(wg, 3), wgrad_TF=round(fl / wg))), flush=True)
    del A, a, w, g, out, dA
