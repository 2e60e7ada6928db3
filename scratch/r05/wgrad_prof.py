"""One shape of ver_wgrad_tn and the library's batched T x N form under rocprofv3 (kernel trace or PMC):
    python3 scratch/r05/wgrad_prof.py <flags> [splits] [reps]     (layer 3, class (0,0): 345 600 x 14 304 x 1 536)"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hip = importlib.import_module('vln-ver_amd.hipops')
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
M, Ka, ld, N = 345600, 14304, 14464, 1536
Af = torch.randn(M, ld, device='cuda', dtype=torch.bfloat16)
G = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
A = Af[:, :Ka]
for _ in range(reps):
    if flags >= 0:
        hip.wgrad_tn(A, G, splits=splits, flags=flags)
    else:
        s = 8
        torch.bmm(A.unflatten(0, (s, M // s)).transpose(1, 2), G.unflatten(0, (s, M // s))).sum(0, dtype=torch.float32).to(A.dtype)
torch.cuda.synchronize()
