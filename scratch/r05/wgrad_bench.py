"""ver_wgrad_tn (csrc/ver_wgrad.hip) against the library's batched T x N GEMM (rows_tn's round-4 form) on the weight-gradient
shapes of the 192-viewpoint step, plus small-shape correctness against an fp32 product.
    python scratch/r05/wgrad_bench.py [check] [big] [sweep]"""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
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


def lib_tn(A, G, s=8):
    M = A.shape[0]
    a3 = A.unflatten(0, (s, M // s)); g3 = G.unflatten(0, (s, M // s))
    return torch.bmm(a3.transpose(1, 2), g3).sum(0, dtype=torch.float32).to(A.dtype)


if 'check' in what:
    torch.manual_seed(0)
    for (M, Ka, ld, c0, N, ldg, S) in [(2048, 256, 256, 0, 256, 256, 1), (4096, 480, 640, 64, 384, 384, 2),
                                       (16384, 224, 1024, 128, 260, 264, 8), (32768, 1000, 1024, 0, 772, 776, 16),
                                       (128 * 40, 256, 256, 0, 256, 256, 8), (1800, 300, 304, 0, 132, 136, 0), (14400 + 7, 520, 520, 0, 256, 256, 3),
                                       (50000, 64, 64, 0, 64, 64, 0)]:
        Af = torch.randn(M, ld, device=dev).to(torch.bfloat16)
        Gf = torch.randn(M, ldg, device=dev).to(torch.bfloat16)
        A, G = Af[:, c0:c0 + Ka], Gf[:, :N]
        want = A.float().t() @ G.float()
        for flags in (0, 6, 10, 12):
            got = hip.wgrad_tn(A, G, out_dtype=torch.float32, splits=S, flags=flags)
            err = float((got - want).abs().max()); ref = float(want.abs().max())
            gotb = hip.wgrad_tn(A, G, splits=S, flags=flags).float()
            errb = float((gotb - want).norm() / want.norm())
            print('check M=%d Ka=%d ld=%d c0=%d N=%d S=%d pf=%d: max|d| %.3e of %.1f, bf16 rel-L2 %.2e %s'
                  % (M, Ka, ld, c0, N, S, flags, err, ref, errb, 'OK' if err < 2e-3 * ref and errb < 4e-3 else 'FAIL'), flush=True)
        # default splits
        got = hip.wgrad_tn(A, G, out_dtype=torch.float32)
        print('   auto splits: max|d| %.3e' % float((got - want).abs().max()), flush=True)

SHAPES = [('L3 c00', 345600, 14304, 14464, 0, 1536), ('L3 c10', 345600, 9536, 14464, 768, 1536), ('L2 c00', 86400, 14304, 14464, 0, 1536),
          ('L1', 86400, 38400, 38400, 0, 1536), ('occ_proj g0', 552960, 832, 832, 0, 4480)]
if 'big' in what or 'sweep' in what:
    res = []
    for name, M, Ka, ld, c0, N in SHAPES:
        if 'sweep' in what and name != 'L3 c00':
            continue
        Af = torch.randn(M, ld, device=dev, dtype=torch.bfloat16)
        G = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        A = Af[:, c0:c0 + Ka]
        gf = 2.0 * M * Ka * N / 1e9
        ms0 = timeit(lambda: lib_tn(A, G))
        ref = lib_tn(A, G).float()
        row = dict(shape=name, M=M, Ka=Ka, N=N, gflop=gf, lib_ms=round(ms0, 3), lib_tf=round(gf / ms0, 1))
        print(name, 'library bmm x8: %.3f ms = %.0f TFLOP/s' % (ms0, gf / ms0), flush=True)
        combos = [(0, 0)] if 'sweep' not in what else [(s, f) for s in (8, 16) for f in (0, 5, 10, 12)]
        for s, f in combos:
            try:
                ms = timeit(lambda: hip.wgrad_tn(A, G, splits=s, flags=f))
            except Exception as ex:
                print('  fail', s, f, ex); continue
            got = hip.wgrad_tn(A, G, splits=s, flags=f).float()
            rel = float((got - ref).norm() / ref.norm())
            row['ours_s%d_pf%d_ms' % (s, f)] = round(ms, 3)
            print('  ver_wgrad_tn splits=%d pf=%d: %.3f ms = %.0f TFLOP/s (incl. reduce), rel-L2 vs library %.2e'
                  % (s, f, ms, gf / ms, rel), flush=True)
        res.append(row)
        del Af, A, G
    print(json.dumps(res))
