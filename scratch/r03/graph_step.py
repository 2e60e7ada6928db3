"""Can one lifting train step (fwd + bwd + clip + AdamW) be captured in a hipGraph?  Usage: graph_step.py <B> [fwd|step]"""
import importlib, sys, time, traceback, types
import numpy as np, torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
what = sys.argv[2] if len(sys.argv) > 2 else 'step'
dev = torch.device('cuda', 0)
args = types.SimpleNamespace(config=None, workload='vocc_c2f_train', dtype='bf16', micro=192)
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.LiftTrainer(head, 192, 'bf16').to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True, capturable=True)
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
def step():
    loss = model(feats, w2p, org, gt)
    if what == 'fwd':
        return loss
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 300.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print('eager: %.2f ms per %s, loss %.5f' % (timeit(step), what, float(step())), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        static_loss = step()
    torch.cuda.synchronize()
    print('captured; replay: %.2f ms per %s, loss %.5f' % (timeit(g.replay), what, float(static_loss)), flush=True)
except Exception:
    traceback.print_exc()
    print('CAPTURE FAILED')
