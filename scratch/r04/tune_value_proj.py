"""TunableOp pass over the head-major value_proj (batched GEMM over the 8 heads, N = 96) for the batch sizes of bench.py:
appends the chosen hipBLASLt solutions to gpurun_out/r04_tunable_vp0.csv (merge into vln-ver_amd/tuning/)."""
import torch, json, os, shutil, importlib, sys
import torch.cuda.tunable as tunable
sys.path.insert(0, '.')
dev = 'cuda'
shutil.copy('vln-ver_amd/tuning/tunableop_gfx950_vocc.csv', 'gpurun_out/r04_tunable_vp0.csv')
tunable.enable(True)
tunable.set_filename('gpurun_out/r04_tunable_vp.csv', True)
tunable.read_file('gpurun_out/r04_tunable_vp0.csv')
tunable.set_max_tuning_duration(30)
tunable.tuning_enable(True)
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H, C = 8, 768
for B in (192, 64, 8, 2, 1):
    M = B * 6 * 196
    x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    W = torch.randn(C, C, device=dev, dtype=torch.bfloat16) * 0.03
    b = torch.randn(C, device=dev, dtype=torch.bfloat16)
    f1 = lambda: torch.baddbmm(b.view(H, 1, C // H), x.unsqueeze(0).expand(H, M, C), W.view(H, C // H, C).transpose(1, 2))
    f0 = lambda: torch.addmm(b, x, W.t())
    print(json.dumps(dict(B=B, M=M, baddbmm_tuned_ms=timeit(f1), plain_ms=timeit(f0))), flush=True)
tunable.write_file()
print(open('gpurun_out/r04_tunable_vp0.csv').read()[-1500:])
