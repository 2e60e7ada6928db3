"""ver_dgrad_nt (csrc/ver_gemm.hip) against the library's mm + 3 x addmm(beta = 1) on the d(tap matrix) shapes of the step.
    python scratch/r05/dgrad_bench.py [check] [big]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
dev = 'cuda'
what = sys.argv[1:] or ['check', 'big']


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def lib(gs, ws, c0s, d_a, kt):
    d_a[:, c0s[0] + ws[0].shape[0]:] = 0
    for i, (g, w, c0) in enumerate(zip(gs, ws, c0s)):
        c1 = c0 + w.shape[0]
        if i == 0:
            torch.mm(g, w.t(), out=d_a[:, c0:c1])
        else:
            torch.addmm(d_a[:, c0:c1], g, w.t(), out=d_a[:, c0:c1])
    return d_a


if 'check' in what:
    torch.manual_seed(0)
    for (M, N, kt, segs) in [(256, 128, 256, [(0, 256)]), (300, 256, 700, [(0, 640), (100, 300), (520, 180)]),
                             (1000, 128, 1000, [(8, 900), (8, 40), (300, 700), (992, 8)]), (513, 384, 520, [(16, 100), (200, 64)])]:
        gs = [torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in segs]
        ws = [torch.randn(kc, N, device=dev).to(torch.bfloat16) for _, kc in segs]
        c0s = [c0 for c0, _ in segs]
        want = torch.zeros(M, kt, device=dev)
        for g, w, c0 in zip(gs, ws, c0s):
            want[:, c0:c0 + w.shape[0]] += g.float() @ w.float().t()
        out = torch.full((M, kt + 8), 5.0, device=dev, dtype=torch.bfloat16)
        hip.dgrad_nt(gs, ws, c0s, out[:, :kt])
        err = float((out[:, :kt].float() - want).norm() / want.norm())
        print('check M=%d N=%d Kt=%d segs=%s: rel-L2 %.2e, untouched pad %s, uncovered zero %s %s' % (
            M, N, kt, segs, err, float(out[:, kt:].min()) == 5.0, bool((out[:, :kt].float()[:, (want.abs().sum(0) == 0)] == 0).all()),
            'OK' if err < 4e-3 else 'FAIL'), flush=True)

if 'big' in what:
    # layer 3 / 2 (N = 1536, kt = 14464): class ranges of dense_heads/upsample.py::_layer_plan_z4 for ci = 768
    up = importlib.import_module('vln-ver_amd.dense_heads.upsample')
    plan, kt, total_rows, taps, offs = up._layer_plan_z4(768, torch.device(dev))
    rng = [(plan[c][0], plan[c][1] - plan[c][0]) for c in up._CLASSES]
    print('class ranges', rng, 'kt', kt)
    for name, M in (('L3', 345600), ('L2', 86400), ('L3 B=64', 115200)):
        gs = [torch.randn(M, 1536, device=dev, dtype=torch.bfloat16) for _ in rng]
        ws = [torch.randn(kc, 1536, device=dev, dtype=torch.bfloat16) for _, kc in rng]
        c0s = [c0 for c0, _ in rng]
        d_a = torch.empty(M, kt, device=dev, dtype=torch.bfloat16)
        gf = sum(2.0 * M * kc * 1536 for _, kc in rng) / 1e9
        ms0 = timeit(lambda: lib(gs, ws, c0s, d_a, kt))
        ref = d_a[-20000:].float().clone()
        d_a.zero_()
        ms = timeit(lambda: hip.dgrad_nt(gs, ws, c0s, d_a))
        rel = float((d_a[-20000:].float() - ref).norm() / ref.norm())
        print('%s: library mm + 3 addmm %.3f ms = %.0f TFLOP/s | ver_dgrad_nt %.3f ms = %.0f TFLOP/s | rel-L2 vs library %.2e' % (
            name, ms0, gf / ms0, ms, gf / ms, rel), flush=True)
        del gs, ws, d_a
    # layer 1: one class, Kc = 38400
    M = 86400
    g = torch.randn(M, 1536, device=dev, dtype=torch.bfloat16); w = torch.randn(38400, 1536, device=dev, dtype=torch.bfloat16)
    d_a = torch.empty(M, 38400, device=dev, dtype=torch.bfloat16)
    gf = 2.0 * M * 38400 * 1536 / 1e9
    ms0 = timeit(lambda: torch.mm(g, w.t(), out=d_a)); ref = d_a[-20000:].float().clone(); d_a.zero_()
    ms = timeit(lambda: hip.dgrad_nt([g], [w], [0], d_a))
    print('L1: library %.3f ms = %.0f TFLOP/s | ver_dgrad_nt %.3f ms = %.0f TFLOP/s | rel-L2 %.2e' % (
        ms0, gf / ms0, ms, gf / ms, float((d_a[-20000:].float() - ref).norm() / ref.norm())), flush=True)
