"""Weight-gradient GEMMs of the step (GEMM ledger: 81 ms at 0.35-0.45 of the MFMA peak) in other operand layouts.
    dW = A^T G,  A [M, K] (column range of a wider row-major matrix), G [M, N] row-major, M >> K, N.
  v0  bmm over 8 row chunks of (A^T view, G) -- what rows_tn() does today (torch T x N)
  v1  (G^T copy [N, M]) x A -> dW^T          (torch N x N: the forward GEMMs' layout class), then transpose the small result
  v1b the same over 8 row chunks as one bmm
  v2  mm(A^T view, G) unsplit
Each with the library default and with TunableOp tuning (15 ms per candidate)."""
import sys, os, time, json
import torch
import torch.cuda.tunable as tunable
dev = 'cuda'
SHAPES = [  # (name, M, K, ldA, N)
    ('L3 c00', 345600, 14304, 14464, 1536),
    ('L3 c10', 345600, 9536, 14464, 1536),
    ('L3 c01', 345600, 6304, 14464, 1536),
    ('L2 c00', 86400, 14304, 14464, 1536),
    ('L1', 86400, 38400, 38400, 1536),
    ('occ_proj g0', 552960, 824, 824, 4480),
    ('occ_proj g1', 552960, 728, 728, 4480),
]
only = sys.argv[1:] 
def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
TUNE = os.environ.get('WGRAD_TUNE', '0') == '1'     # (tuning a 15-TFLOP GEMM takes minutes per variant: off by default)
if TUNE:
    tunable.enable(True)
    tunable.set_max_tuning_duration(15)
    tunable.set_filename('gpurun_out/r04_wgrad_tunable.csv', True)
res = []
for name, M, K, ld, N in SHAPES:
    if only and not any(o in name for o in only): continue
    A_full = torch.randn(M, ld, device=dev, dtype=torch.bfloat16)
    A = A_full[:, :K]
    G = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    s = 8
    def v0():
        a3 = A.unflatten(0, (s, M // s)); g3 = G.unflatten(0, (s, M // s))
        return torch.bmm(a3.transpose(1, 2), g3).sum(0, dtype=torch.float32).to(A.dtype)
    def v1():
        gT = G.t().contiguous()
        return torch.mm(gT, A).t()
    def v1b():
        g3 = G.unflatten(0, (s, M // s)).transpose(1, 2).contiguous()          # [s, N, M/s]
        a3 = A.unflatten(0, (s, M // s))
        return torch.bmm(g3, a3).sum(0, dtype=torch.float32).to(A.dtype).t()
    def v2():
        return torch.mm(A.t(), G)
    def tr_only():
        return G.t().contiguous()
    ref = None
    row = dict(shape=name, M=M, K=K, N=N, gflop=2.0 * M * K * N / 1e9)
    for tuned in ((False, True) if TUNE else (False,)):
        if TUNE: tunable.tuning_enable(tuned)
        for nm, fn in (('v0', v0), ('v1', v1), ('v1b', v1b), ('v2', v2), ('transpose', tr_only)):
            try:
                ms = timeit(fn)
            except Exception as ex:
                ms = float('nan'); print('fail', nm, ex)
            row['%s_%s' % (nm, 'tuned' if tuned else 'default')] = round(ms, 3)
            print(name, nm, 'tuned' if tuned else 'default', round(ms, 3), 'ms', round(row['gflop'] / ms, 0), 'TF', flush=True)
    out = v0().float(); alt = v1().float()
    row['v1_vs_v0_rel'] = float((out - alt).norm() / out.norm())
    res.append(row)
    print(json.dumps(row), flush=True)
    del A_full, A, G
if TUNE: tunable.write_file()
