"""Forward product of the lattice layers at B viewpoints: ver_lattice_gather + ver_gemm_nn against ver_gemm_nn_taps."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
ups = importlib.import_module('vln-ver_amd.dense_heads.upsample')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
DEV, C, N = 'cuda', 768, 1536


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


plan, kt, _, taps18, offs18 = ups._layer_plan_z4(C, torch.device(DEV))
for name, layout, hc in (('layer 2', 2, 15), ('layer 3', 3, 30)):
    m = B * 2 * hc * hc
    shape = (4, B, 2, hc // 2, hc // 2, 2, C) if layout == 3 else (B, 2, hc, hc, 2, C)
    e = torch.randn(shape, device=DEV).bfloat16()
    a = torch.empty(m, kt, dtype=torch.bfloat16, device=DEV)
    t_g = timeit(lambda: ups._gather_z4(e, layout, a, taps18, offs18, C, hc, hc, with_const=True))
    tot_e = tot_i = 0.0
    for cls in ups._CLASSES:
        c0, c1 = plan[cls][:2]
        # the class's tap blocks, in column order (constant blocks skipped: they ride in the position table)
        blocks = [t for t in range(18) if c0 <= ups._block_offset4(t, C) < c1]
        sub = [taps18[t] for t in blocks]
        w_e = (torch.randn(c1 - c0, N, device=DEV) * 0.05).bfloat16()
        w_i = (torch.randn(len(sub) * C, N, device=DEV) * 0.05).bfloat16()
        rowpos = torch.randn(2 * hc * hc, N, device=DEV)
        out = torch.empty(m, N, dtype=torch.bfloat16, device=DEV)
        t_e = timeit(lambda: hip.gemm_nn(a[:, c0:c1], w_e, out=out, splits=1))
        t_i = timeit(lambda: hip.gemm_nn_taps(e, layout, (hc, hc), sub, w_i, rowpos=rowpos, out=out))
        table, _ = ups._const_rows_z4(C, hc, hc, torch.device(DEV), torch.bfloat16)
        segs = ups._class_segments_z4(cls, C)
        t_s = timeit(lambda: hip.gemm_nn_taps(e, layout, (hc, hc), segs, w_e, const_rows=table, out=out))
        fl_e, fl_i = 2.0 * m * (c1 - c0) * N, 2.0 * m * len(sub) * C * N
        print('%s class %s: explicit %.3f ms (%.0f TFLOP/s, K = %d)   implicit + position table %.3f ms (%.0f TFLOP/s, K = %d)   implicit '
              'with pattern segments %.3f ms (%.0f TFLOP/s)' % (name, cls, t_e, fl_e / t_e / 1e9, c1 - c0, t_i, fl_i / t_i / 1e9, len(sub) * C,
                                                                 t_s, fl_e / t_s / 1e9), flush=True)
        tot_e += t_e; tot_i += t_s
    print('%s: gather %.3f ms + explicit GEMMs %.3f ms = %.3f ms;  implicit GEMMs %.3f ms' % (name, t_g, tot_e, t_g + tot_e, tot_i), flush=True)
    del e, a
# layer 1: plain source, 50 taps, no classes
hc = 15
m = B * 2 * hc * hc
taps, offs, lo, hi = ups._layer0_z4_plan(C, torch.device(DEV))
e = torch.randn(B, 4, hc, hc, C, device=DEV).bfloat16()
a = torch.empty(m, 50 * C, dtype=torch.bfloat16, device=DEV)
t_g = timeit(lambda: ups._gather_z4(e, 0, a, taps, offs, C, hc, hc))
w = (torch.randn(50 * C, N, device=DEV) * 0.05).bfloat16()
bias = torch.randn(N, device=DEV)
out = torch.empty(m, N, dtype=torch.bfloat16, device=DEV)
t_e = timeit(lambda: hip.gemm_nn(a, w, bias, out=out, splits=1))
t_i = timeit(lambda: hip.gemm_nn_taps(e, 0, (hc, hc), taps, w, bias=bias, out=out))
print('layer 1: gather %.3f ms + explicit GEMM %.3f ms = %.3f ms;  implicit GEMM %.3f ms (%.0f TFLOP/s)' % (t_g, t_e, t_g + t_e, t_i, 2.0 * m * 50 * C * N / t_i / 1e9))

# ---- weight gradient: ver_wgrad_tn on the explicit tap matrix against ver_wgrad_tn_segments
print('--- weight gradient')
for name, layout, hc in (('layer 2', 2, 15), ('layer 3', 3, 30)):
    m = B * 2 * hc * hc
    shape = (4, B, 2, hc // 2, hc // 2, 2, C) if layout == 3 else (B, 2, hc, hc, 2, C)
    e = torch.randn(shape, device=DEV).bfloat16()
    a = torch.empty(m, kt, dtype=torch.bfloat16, device=DEV)
    ups._gather_z4(e, layout, a, taps18, offs18, C, hc, hc, with_const=True)
    table, _ = ups._const_rows_z4(C, hc, hc, torch.device(DEV), torch.bfloat16)
    g = torch.randn(m, N, device=DEV).bfloat16()
    tot_e = tot_i = 0.0
    for cls in ups._CLASSES:
        c0, c1 = plan[cls][:2]
        segs = ups._class_segments_z4(cls, C)
        out = torch.empty(c1 - c0, N, dtype=torch.bfloat16, device=DEV)
        t_e = timeit(lambda: hip.wgrad_tn(a[:, c0:c1], g, out=out))
        t_i = timeit(lambda: hip.wgrad_tn_segments(e, layout, (hc, hc), segs, g, out=out, const_rows=table))
        fl = 2.0 * m * (c1 - c0) * N
        print('%s class %s: explicit %.3f ms (%.0f TFLOP/s)   implicit %.3f ms (%.0f TFLOP/s)' % (name, cls, t_e, fl / t_e / 1e9, t_i, fl / t_i / 1e9), flush=True)
        tot_e += t_e; tot_i += t_i
    print('%s: explicit %.3f ms, implicit %.3f ms' % (name, tot_e, tot_i), flush=True)
    del e, a, g
hc = 15
m = B * 2 * hc * hc
taps, offs, lo, hi = ups._layer0_z4_plan(C, torch.device(DEV))
e = torch.randn(B, 4, hc, hc, C, device=DEV).bfloat16()
a = torch.empty(m, 50 * C, dtype=torch.bfloat16, device=DEV)
ups._gather_z4(e, 0, a, taps, offs, C, hc, hc)
g = torch.randn(m, N, device=DEV).bfloat16()
out = torch.empty(50 * C, N, dtype=torch.bfloat16, device=DEV)
t_e = timeit(lambda: hip.wgrad_tn(a, g, out=out))
t_i = timeit(lambda: hip.wgrad_tn_segments(e, 0, (hc, hc), taps, g, out=out))
print('layer 1: explicit %.3f ms (%.0f TFLOP/s), implicit %.3f ms (%.0f TFLOP/s)' % (t_e, 2.0 * m * 50 * C * N / t_e / 1e9, t_i, 2.0 * m * 50 * C * N / t_i / 1e9))
