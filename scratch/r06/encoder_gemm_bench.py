"""The encoder's tall Linears of the 192-viewpoint step on ver_gemm_nn against the library (tuned table): forward x @ W^T + b and
d(input) g @ W.    python scratch/r06/encoder_gemm_bench.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
dev = 'cuda'
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, K, N in [(172800, 768, 1536), (172800, 1536, 768), (225792, 768, 768), (172800, 768, 768), (172800, 768, 192), (172800, 192, 768)]:
    x = torch.relu(torch.randn(M, K, device=dev)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)           # nn.Linear weight [out, in]
    b = torch.randn(N, device=dev)
    wt = w.t().contiguous()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms0 = timeit(lambda: torch.addmm(b.to(torch.bfloat16), x, w.t(), out=out))
    ref = out.float().clone()
    ms1 = timeit(lambda: hip.gemm_nn(x, wt, b, out=out))
    rel = float((out.float() - ref).norm() / ref.norm())
    g = torch.randn(M, N, device=dev).to(torch.bfloat16)
    gx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    ms2 = timeit(lambda: torch.mm(g, w, out=gx))
    ref = gx.float().clone()
    ms3 = timeit(lambda: hip.gemm_nn(g, w, out=gx))
    rel2 = float((gx.float() - ref).norm() / ref.norm())
    tf = 2.0 * M * K * N / 1e9
    print('[%d, %d] -> %d: forward library %.0f us (%4.0f TF/s) ours %.0f us (%4.0f) rel %.0e | d(input) library %.0f us (%4.0f) ours %.0f us (%4.0f) rel %.0e'
          % (M, K, N, ms0 * 1e3, tf / ms0, ms1 * 1e3, tf / ms1, rel, ms2 * 1e3, tf / ms2, ms3 * 1e3, tf / ms3, rel2), flush=True)
