"""Do memset nodes / torch's multi-block reductions survive hipGraph replay on this stack?"""
import torch
dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(900, 768, device=dev).bfloat16()
z = torch.empty(1 << 16, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = x.sum(0); z.zero_(); z.add_(1.0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x.sum(0)
    y32 = x.float().sum(0)
    z.zero_()              # (hipMemsetAsync -> a memset node)
    z.add_(1.0)
    w = torch.zeros(4096, device=dev); w.add_(2.0)
for i in range(6):
    x.normal_()
    z.fill_(float('nan'))
    g.replay(); torch.cuda.synchronize()
    ref = x.sum(0)
    print('replay %d: bf16 colsum max |d| %.3e (finite %s), fp32 colsum finite %s, z == 1: %s, w == 2: %s'
          % (i, float((y.float() - ref.float()).abs().max()), bool(torch.isfinite(y).all()), bool(torch.isfinite(y32).all()),
             bool((z == 1).all()), bool((w == 2).all())), flush=True)
