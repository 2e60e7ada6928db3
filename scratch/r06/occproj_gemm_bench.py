"""occ_proj's four products of the 192-viewpoint step on ver_gemm_nn against the library (tuned table): forward [552960, k] x [k, 4480]
and d(input) [552960, 4480] x [4480, n] for k / n = 832, 768 (+ the d(input) widths that are actually needed: 820 / 772 / 724).
    python scratch/r06/occproj_gemm_bench.py"""
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
M = 552960
for name, K, N, ldw in [('fwd k=832', 832, 4480, 4480), ('fwd k=768', 768, 4480, 4480), ('dgrad n=832', 4480, 832, 832), ('dgrad n=768', 4480, 768, 768),
                        ('dgrad n=820 of 832', 4480, 820, 832), ('dgrad n=772 of 832', 4480, 772, 832), ('dgrad n=724 of 768', 4480, 724, 768)]:
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    Wf = (torch.randn(K, ldw, device=dev) * 0.05).to(torch.bfloat16)
    W = Wf[:, :N]
    outf = torch.empty(M, ldw if N != ldw else N, device=dev, dtype=torch.bfloat16)
    out = outf[:, :N]
    tf = 2.0 * M * K * N / 1e12
    if N == ldw:
        ms0 = timeit(lambda: torch.mm(A, W, out=out))
        ref = out[-30000:].float().clone()
    else:
        full = torch.empty(M, ldw, device=dev, dtype=torch.bfloat16)
        ms0 = timeit(lambda: torch.mm(A, Wf, out=full))
        ref = full[-30000:, :N].float().clone(); del full
    outf.zero_()
    ms = timeit(lambda: hip.gemm_nn(A, W, out=out))
    got = out[-30000:].float()
    rel = float((got - ref).norm() / ref.norm())
    print('%-20s library %.3f ms (%4.0f TFLOP/s%s)  ver_gemm_nn %.3f ms (%4.0f TFLOP/s)  rel-L2 of the last rows %.1e'
          % (name, ms0, 2.0 * M * K * ldw / 1e9 / ms0, '' if N == ldw else ', full width', ms, tf * 1e3 / ms, rel), flush=True)
    del A, Wf, outf
