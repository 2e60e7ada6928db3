"""Timing of the fused focal forward+grad pass (ver_focal_loss_forward_grad, in place) at the step's size."""
import importlib, sys, ctypes, os
sys.path.insert(0, '.')
import torch
hip = importlib.import_module('vln-ver_amd.hipops')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96768000
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
l2 = torch.randn(N, 16, device=dev, generator=g).to(torch.bfloat16)
tgt = torch.randint(0, 17, (N,), device=dev, generator=g)
blocks = hip.lib().ver_focal_loss_blocks(ctypes.c_long(N), 16)
partial = torch.zeros(blocks, dtype=torch.float32, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
ref = l2[:4096].clone()
def run(buf):
    rc = hip.lib().ver_focal_loss_forward_grad(hip._p(buf), hip._p(tgt), hip._p(partial), hip._p(buf), ctypes.c_long(N), 16, ctypes.c_float(2.0),
                                             ctypes.c_float(0.25), 1, hip._p(flag), hip._stream())
    assert rc == 0
work = l2.clone()
for _ in range(2): run(work)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run(work)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print('focal fwd+grad N=%d: %.3f ms, %.2f TB/s (2*N*32 B + N*8 B)' % (N, ms, (N * 72) / ms / 1e9))
# value check of one pass against torch on a slice
work = l2.clone(); run(work); torch.cuda.synchronize()
x = ref.float(); t = tgt[:4096]
oh = torch.nn.functional.one_hot(t, 17)[:, :16].float()
p = torch.sigmoid(x); pt = (1 - p) * oh + p * (1 - oh)
fw = (0.25 * oh + 0.75 * (1 - oh)) * pt.pow(2)
loss = torch.nn.functional.binary_cross_entropy_with_logits(x, oh, reduction='none') * fw
xx = x.clone().requires_grad_(True)
p2 = torch.sigmoid(xx); pt2 = (1 - p2) * oh + p2 * (1 - oh)
l = (torch.nn.functional.binary_cross_entropy_with_logits(xx, oh, reduction='none') * (0.25 * oh + 0.75 * (1 - oh)) * pt2.pow(2)).sum()
l.backward()
print('grad maxdiff vs torch (bf16 store):', float((work[:4096].float() - xx.grad).abs().max()), 'sum rel', float(abs(partial.sum() - 0) > 0))
