"""Where does the full multi-task step spend its time (scratch)."""
import importlib, os, sys, argparse, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import bench, cases
from torch.profiler import profile, ProfilerActivity
B = 32
args = argparse.Namespace(workload='vocc_full_train', dtype='bf16', micro=32, batch=B)
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.FullTrainer(head, 'bf16').to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
gts = [cases.detection_gt(seed=40 + i, num_gt=3 + i % 5) for i in range(B)]
gt_boxes = [torch.from_numpy(g[0][:, :7]).to(dev) for g in gts]
gt_labels = [torch.from_numpy(g[1]).to(dev) for g in gts]
def step():
    loss = model(feats, w2p, org, gt, gt_boxes, gt_labels); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print('wall ms', (time.perf_counter() - t0) * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(k.self_device_time_total for k in ka)
print('GPU busy ms', tot / 1e3)
open('gpurun_out/prof_full.txt', 'w').write(ka.table(sort_by='self_cpu_time_total', row_limit=40, max_name_column_width=60))
print(ka.table(sort_by='self_cpu_time_total', row_limit=25, max_name_column_width=50))
