"""Per-kernel time of the backward gather (torch profiler), 64 viewpoints."""
import sys, os, importlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch, cases
hip = importlib.import_module('vln-ver_amd.hipops'); syn = importlib.import_module('vln-ver_amd.synthetic')
if os.environ.get('VER_LIB'): hip.LIB_PATH = os.environ['VER_LIB']
B = 64; dev = 'cuda'
w2p, org = syn.camera_batch(B, seed=1)
hit = hip.project_points(torch.from_numpy(w2p).to(dev), torch.from_numpy(org).to(dev), cases.PC_RANGE, 4, 15, 15)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, 6, 196, 8, 96, device=dev, generator=g).requires_grad_(True)
offs = (torch.randn(B, 900, 8, 8, 2, device=dev, generator=g) * 3).requires_grad_(True)
logits = torch.randn(B, 900, 8, 8, device=dev, generator=g).requires_grad_(True)
gs = torch.randn(B, 900, 768, device=dev, generator=g)
s = hip.sca_gather(value, offs, logits, hit, 14, 14)
for _ in range(3): torch.autograd.grad(s, [value, offs, logits], gs, retain_graph=True)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(8): torch.autograd.grad(s, [value, offs, logits], gs, retain_graph=True)
    torch.cuda.synchronize()
for k in prof.key_averages():
    if 'k_sca' in k.key or 'fill' in k.key.lower(): print(k.key[:48], round(k.device_time_total / k.count, 1), 'us x', k.count)
