"""ver_gemm_nn (csrc/ver_gemm.hip) against the library's GEMM on the forward shapes of the 192-viewpoint step + small-shape checks.
    python scratch/r05/gemm_bench.py [check] [big]"""
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


if 'check' in what:
    torch.manual_seed(0)
    for (M, K, lda, c0, N, ldw, bias) in [(256, 64, 64, 0, 256, 256, False), (512, 128, 192, 64, 256, 256, True), (300, 96, 96, 0, 200, 200, True),
                                          (1000, 1024, 1088, 32, 772, 776, False), (4096, 2048, 2048, 0, 1536, 1536, False), (77, 320, 320, 0, 40, 40, True)]:
        Af = torch.randn(M, lda, device=dev).to(torch.bfloat16)
        Wf = torch.randn(K, ldw, device=dev).to(torch.bfloat16)
        b = torch.randn(N, device=dev) if bias else None
        A, W = Af[:, c0:c0 + K], Wf[:, :N]
        want = A.float() @ W.float() + (b if bias else 0)
        got = hip.gemm_nn(A, W, b).float()
        err = float((got - want).norm() / want.norm()); mx = float((got - want).abs().max())
        # into a column range of a wider output
        wide = torch.full((M, N + 24), 7.0, device=dev, dtype=torch.bfloat16)
        hip.gemm_nn(A, W, b, out=wide[:, 8:8 + N])
        ok2 = bool(torch.equal(wide[:, 8:8 + N].float(), got)) and float(wide[:, :8].min()) == 7.0 and float(wide[:, 8 + N:].min()) == 7.0
        print('check M=%d K=%d lda=%d c0=%d N=%d bias=%s: rel-L2 %.2e max|d| %.3e of %.1f %s %s'
              % (M, K, lda, c0, N, bias, err, mx, float(want.abs().max()), 'OK' if err < 4e-3 else 'FAIL', 'OK' if ok2 else 'FAIL-range'), flush=True)

SHAPES = [('L3 c00', 345600, 14304, 14464, 1536), ('L3 c10', 345600, 9536, 14464, 1536), ('L2 c00', 86400, 14304, 14464, 1536),
          ('L1', 86400, 38400, 38400, 1536), ('occ_proj g0', 552960, 832, 832, 4480), ('L3 c00 B=8', 14400, 14304, 14464, 1536)]
if 'big' in what:
    res = []
    for name, M, K, ld, N in SHAPES:
        Af = torch.randn(M, ld, device=dev, dtype=torch.bfloat16)
        W = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        A = Af[:, :K]
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        gf = 2.0 * M * K * N / 1e9
        ms0 = timeit(lambda: torch.mm(A, W, out=out))
        sl = slice(M - 40000, M) if M > 100000 else slice(0, M)          # (the LAST rows: partial tiles, range ends)
        ref = out[sl].float().clone()
        out.zero_()
        ms = timeit(lambda: hip.gemm_nn(A, W, out=out))
        got = out[sl].float()
        rel = float((got - ref).norm() / ref.norm())
        nz = float((out[:4096].float() == 0).float().mean())
        print('   zero fraction of the first rows after ver_gemm_nn: %.2e; identical elements: %.6f' % (nz, float((got == ref).float().mean())))
        print('%s: library %.3f ms = %.0f TFLOP/s | ver_gemm_nn %.3f ms = %.0f TFLOP/s | rel-L2 %.2e' % (name, ms0, gf / ms0, ms, gf / ms, rel), flush=True)
        res.append(dict(shape=name, M=M, K=K, N=N, lib_ms=round(ms0, 3), ours_ms=round(ms, 3)))
        del Af, A, W, out
    print(json.dumps(res))
