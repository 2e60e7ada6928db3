"""Would d(input) as ver_gemm_nn pay?  d_a[M, Kc] = g[M, 1536] @ Wt[1536, Kc] (Wt = the class weight transposed once per step: B operand
in whole 128-byte lines through the transposing path) against the library's mm(g, w.t()), per class, K = 1536 (48 phases per tile)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
dev = 'cuda'
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, Kc in (('L3 c00', 345600, 14304), ('L3 c11', 345600, 6304), ('L2 c00', 86400, 14304), ('L1', 86400, 38400)):
    g = torch.randn(M, 1536, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Kc, 1536, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    out = torch.empty(M, Kc, device=dev, dtype=torch.bfloat16)
    gf = 2.0 * M * Kc * 1536 / 1e9
    ms0 = timeit(lambda: torch.mm(g, w.t(), out=out))
    ref = out[-10000:].float().clone(); out.zero_()
    ms1 = timeit(lambda: hip.gemm_nn(g, wt, out=out))
    rel = float((out[-10000:].float() - ref).norm() / ref.norm())
    print('%s: library mm(g, w.t()) %.3f ms = %.0f TFLOP/s | ver_gemm_nn(g, wt) %.3f ms = %.0f TFLOP/s | rel %.1e' % (name, ms0, gf / ms0, ms1, gf / ms1, rel), flush=True)
    del g, w, wt, out
