"""ver_lattice_rows against ver_lattice_transpose + ver_run_gather / _scatter on the vocc.py lattice (bs viewpoints)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
opl = importlib.import_module('vln-ver_amd.dense_heads.occ_proj_lattice')
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C, Z, Hl, Wl = 768, 4, 60, 60
dev = torch.device('cuda', 0)
plan = opl.get_plan(C, Z, 2 * Hl, 2 * Wl, dev)
row_map, spans, total = opl._row_map_for(plan, bs)
src = torch.randn(4, bs, 2, Hl // 2, Wl // 2, 2, C, device=dev).bfloat16()
buf = torch.empty(total, dtype=torch.bfloat16, device=dev)
back = torch.empty_like(src)
L = C * Z * Hl * Wl
lat = torch.empty(bs, (L + C + 2 + 7) // 8 * 8, dtype=torch.bfloat16, device=dev)
outs = [torch.empty(bs * g.n_rows, g.k_aug, dtype=torch.bfloat16, device=dev) for g in plan.groups]
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
gb = 2 * bs * L * 2 / 1e9
def two_fwd():
    hip.lattice_transpose(src, lat, (Hl, Wl), 3, True)
    for g, o in zip(plan.groups, outs): hip.run_gather(lat, g.run_start, g.aug_idx, o, g.n_rows, g.run_len)
def two_bwd():
    for g, o in zip(plan.groups, outs): hip.run_scatter(o, lat, g.run_start, g.n_rows, g.run_len)
    hip.lattice_transpose(back, lat, (Hl, Wl), 3, False)
for name, f in (('rows fwd', lambda: hip.lattice_rows(src, buf, row_map, (Hl, Wl), 3, True)), ('rows bwd', lambda: hip.lattice_rows(back, buf, row_map, (Hl, Wl), 3, False)),
                ('transpose fwd', lambda: hip.lattice_transpose(src, lat, (Hl, Wl), 3, True)), ('two-pass fwd', two_fwd), ('two-pass bwd', two_bwd)):
    ms = t(f)
    print('%-14s %7.3f ms   %.2f TB/s of %.2f GB' % (name, ms, gb / ms, gb))
