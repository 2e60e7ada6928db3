"""occ_proj GEMMs (forward 552 960 x 4 480 x K, dgrad 552 960 x N x 4 480) at the shipped operand widths and padded to
multiples of 64 / 128: does another width let hipBLASLt leave the stream-K kernel (0.37 of peak)?  TunableOp tuning on."""
import torch, json, sys
import torch.cuda.tunable as tunable
dev = 'cuda'
tunable.enable(True); tunable.set_max_tuning_duration(20); tunable.tuning_enable(True)
tunable.set_filename('gpurun_out/r04_occproj_tunable.csv', True)
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, N = 552960, 4480
for K in (824, 832, 896, 728, 768, 776):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)      # wa [out, k_aug]; forward = a @ wa.t()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fwd = timeit(lambda: torch.mm(a, w.t(), out=out))
    go = out
    dg = timeit(lambda: torch.mm(go, w))                         # dgrad over all K columns
    fl = 2.0 * M * N * K / 1e9
    print(json.dumps(dict(K=K, fwd_ms=round(fwd, 3), fwd_TF=round(fl / fwd, 0), dgrad_ms=round(dg, 3), dgrad_TF=round(fl / dg, 0))), flush=True)
    del a, w, out
