"""Row-split count of the occ_proj / upsample weight-gradient GEMMs (rows_tn): time per split count, library default vs tuned."""
import os, sys, time, importlib
import torch
sys.path.insert(0, '.')
dev = 'cuda'
def t(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
mode = sys.argv[1] if len(sys.argv) > 1 else 'default'
if mode == 'recorded':
    print('recorded table:', importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms())
elif mode == 'tune':
    import torch.cuda.tunable as tunable
    tunable.enable(True); tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30); tunable.set_max_tuning_iterations(20)
    tunable.set_filename('gpurun_out/tunableop_wgrad_split.csv')
shapes = [(552960, 824, 4480), (552960, 728, 4480), (552960, 776, 4480), (345600, 9536, 1536), (345600, 9376, 1536), (345600, 6304, 1536), (345600, 14304, 1536)]
if len(sys.argv) > 3: shapes = shapes[:int(sys.argv[3])]
splits = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [5, 6, 8, 10, 12, 16, 20, 24, 32]
for (m, k, n) in shapes:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    g = torch.randn(m, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * k * n
    line = 'M=%d K=%d N=%d:' % (m, k, n)
    for s in splits:
        if m % s: continue
        a3 = a.unflatten(0, (s, m // s)); g3 = g.unflatten(0, (s, m // s))
        ms = t(lambda: torch.bmm(a3.transpose(1, 2), g3).sum(0, dtype=torch.float32).to(a.dtype))
        line += ' s=%d %.2f ms (%.2f PF)' % (s, ms, fl / ms / 1e12)
    print(line, flush=True)
    del a, g
