"""Is the head's GEMM rate set by the power limit?  ver_wgrad_tn, ver_gemm_nn and the library's d(input) product on the layer-3
shapes with N(0,1) operands (the bench's data), with operands of one repeated value, and with zeros: the same instruction
stream, different switching activity."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
dev = 'cuda'
M, K, N = 345600, 14304, 1536
def t(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, make in (('N(0,1)', lambda *s: torch.randn(*s, device=dev).bfloat16()),
                   ('constant 0.5', lambda *s: torch.full(s, 0.5, device=dev, dtype=torch.bfloat16)),
                   ('zeros', lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16))):
    a = make(M, K); g = make(M, N); w = make(K, N)
    fl = 2.0 * M * K * N
    ms_w = t(lambda: hip.wgrad_tn(a, g, out_dtype=torch.bfloat16))
    ms_f = t(lambda: hip.gemm_nn(a, w))
    ms_d = t(lambda: torch.mm(g, w.t()))
    print('%-13s ver_wgrad_tn %6.2f ms = %5.0f TFLOP/s   ver_gemm_nn %6.2f ms = %5.0f   library d(input) %6.2f ms = %5.0f' % (
        name, ms_w, fl / ms_w / 1e9, ms_f, fl / ms_f / 1e9, ms_d, fl / ms_d / 1e9))
    del a, g, w
