"""ver_wgrad_tn_segments at layer 3's shapes (192 viewpoints) against ver_wgrad_tn on the explicit tap matrix."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
ups = importlib.import_module('vln-ver_amd.dense_heads.upsample')
B, DEV, C, N = 192, 'cuda', 768, 1536
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
plan, kt, _, taps18, offs18 = ups._layer_plan_z4(C, torch.device(DEV))
layout, hc = 3, 30
m = B * 2 * hc * hc
e = torch.randn(4, B, 2, hc // 2, hc // 2, 2, C, device=DEV).bfloat16()
a = torch.empty(m, kt, dtype=torch.bfloat16, device=DEV)
ups._gather_z4(e, layout, a, taps18, offs18, C, hc, hc, with_const=True)
table, _ = ups._const_rows_z4(C, hc, hc, torch.device(DEV), torch.bfloat16)
g = torch.randn(m, N, device=DEV).bfloat16()
te = ti = 0
for cls in ups._CLASSES:
    c0, c1 = plan[cls][:2]
    segs = ups._class_segments_z4(cls, C)
    out = torch.empty(c1 - c0, N, dtype=torch.bfloat16, device=DEV)
    te += timeit(lambda: hip.wgrad_tn(a[:, c0:c1], g, out=out))
    ti += timeit(lambda: hip.wgrad_tn_segments(e, layout, (hc, hc), segs, g, out=out, const_rows=table))
print('PF=%s layer 3 weight gradient: explicit %.3f ms, implicit %.3f ms' % (os.environ.get('VER_WGRAD_SEG_PF', '3'), te, ti))
