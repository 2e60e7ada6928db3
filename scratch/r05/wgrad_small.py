"""ver_wgrad_tn against torch.mm on the decoder's weight-gradient shapes (rows = 100 queries x viewpoints)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
dev = 'cuda'
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for m, ka, n in [(6400, 768, 768), (6400, 768, 1536), (6400, 1536, 768), (57600, 768, 768), (800, 768, 768), (100, 768, 768), (6400, 768, 96), (6400, 768, 32), (6400, 768, 20)]:
    a = torch.randn(m, ka, device=dev).bfloat16(); g = torch.randn(m, n, device=dev).bfloat16()
    lib_us = t(lambda: g.t() @ a)
    own = hip.wgrad_tn_supported(g, a)
    own_us = t(lambda: hip.wgrad_tn(g, a, out_dtype=torch.bfloat16)) if own else float('nan')
    err = float(((hip.wgrad_tn(g, a, out_dtype=torch.float32) - g.float().t() @ a.float()).norm() / (g.float().t() @ a.float()).norm())) if own else float('nan')
    print('M %6d  %4d x %4d   library %7.1f us   ver_wgrad_tn %7.1f us   rel err %.1e' % (m, n, ka, lib_us, own_us, err))
