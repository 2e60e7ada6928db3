"""value_proj as one [M,768]x[768,768] GEMM (reference layout, rows hold all heads) vs a batched GEMM over the 8 heads
writing the head-major layout [heads, M, 96] (a (camera, head) tile of the gather becomes ONE contiguous 37.6-KB block)."""
import torch, json
dev = 'cuda'
M, C, H = 192 * 6 * 196, 768, 8
x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
W = torch.randn(C, C, device=dev, dtype=torch.bfloat16) * 0.03
b = torch.randn(C, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
Wh = W.view(H, C // H, C)
bh = b.view(H, 1, C // H)
f0 = lambda: torch.addmm(b, x, W.t())
f1 = lambda: torch.baddbmm(bh, x.unsqueeze(0).expand(H, M, C), Wh.transpose(1, 2))
f2 = lambda: torch.stack([torch.addmm(b[h * 96:(h + 1) * 96], x, W[h * 96:(h + 1) * 96].t()) for h in range(H)])
f3 = lambda: torch.addmm(b, x, W.t()).view(M, H, 96).permute(1, 0, 2).contiguous()
r = dict(plain_ms=timeit(f0), baddbmm_expand_ms=timeit(f1), eight_gemms_stack_ms=timeit(f2), plain_plus_permute_ms=timeit(f3))
a, c = f0().view(M, H, 96).permute(1, 0, 2).float(), f1().float()
r['max_diff'] = float((a - c).abs().max())
print(json.dumps(r))
