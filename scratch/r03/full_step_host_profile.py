"""Where the host time of the full multi-task step goes at one viewpoint per step (cProfile of 5 steps + CPU-side torch profiler)."""
import cProfile, importlib, io, os, pstats, sys, time, types
import numpy as np, torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda', 0)
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
args = types.SimpleNamespace(config=None, workload='vocc_full_train', dtype='bf16', micro=192)
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.FullTrainer(head, 'bf16').to(dev).train()
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
gts = [syn.detection_gt(seed=40 + i, num_gt=3 + i % 5) for i in range(B)]
gb = [torch.from_numpy(g[0][:, :7]).to(dev) for g in gts]; gl = [torch.from_numpy(g[1]).to(dev) for g in gts]
loss = model(feats, w2p, org, gt, gb, gl); loss.backward()
for p in model.parameters():
    if p.requires_grad and p.grad is None: p.requires_grad_(False)
    p.grad = None
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
marks = {}
def step(timed=False):
    t0 = time.perf_counter()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        outs = head(feats, None, world2pixel=w2p, origin=org, occupancy_rows=True)
    if timed: torch.cuda.synchronize()
    t1 = time.perf_counter()
    outs = {k: (v.float() if torch.is_tensor(v) and k != 'occupancy_preds' else v) for k, v in outs.items()}
    loss = sum(head.loss(gb, gl, gt, outs).values())
    if timed: torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    if timed: torch.cuda.synchronize()
    t3 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)
    if timed: torch.cuda.synchronize()
    t4 = time.perf_counter()
    if timed:
        for k, v in (('forward', t1 - t0), ('loss', t2 - t1), ('backward', t3 - t2), ('clip+adamw', t4 - t3)):
            marks[k] = marks.get(k, 0.0) + v
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print('B=%d: %.1f ms per step' % (B, (time.perf_counter() - t) / 10 * 1e3))
for _ in range(10): step(True)
print('synchronised phases (ms):', {k: round(v / 10 * 1e3, 2) for k, v in marks.items()})
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
