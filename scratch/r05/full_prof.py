"""Which aten ops (with input shapes) launch the small kernels of the full multi-task step.  Scratch tool."""
import importlib, os, sys, argparse, collections
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--out', default='gpurun_out/full_prof.txt')
a = ap.parse_args()
args = argparse.Namespace(workload="vocc_full_train", dtype="bf16", micro=192, batch=a.batch, config=None)
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.FullTrainer(head, 'bf16').to(dev).train()
B = a.batch
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
gts = [syn.detection_gt(seed=40 + i, num_gt=3 + i % 5) for i in range(B)]
gt_boxes = [torch.from_numpy(g[0][:, :7]).to(dev) for g in gts]
gt_labels = [torch.from_numpy(g[1]).to(dev) for g in gts]
model(feats[:, :2], w2p[:2], org[:2], gt[:2], gt_boxes[:2], gt_labels[:2]).backward()
for prm in model.parameters():
    if prm.requires_grad and prm.grad is None:
        prm.requires_grad_(False)
    prm.grad = None
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)

def step():
    loss = model(feats, w2p, org, gt, gt_boxes, gt_labels); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)

for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
with open(a.out, 'w') as f:
    f.write(ka.table(sort_by='self_device_time_total', row_limit=500, max_name_column_width=60, max_shapes_column_width=120))
    f.write('\n\n==== by count ====\n')
    rows = [(e.count, e.key, str(e.input_shapes)[:150], e.self_device_time_total) for e in ka if e.self_device_time_total > 0 or e.key.startswith('aten::')]
    rows.sort(key=lambda r: -r[0])
    for r in rows[:250]:
        f.write('%6d %-45s %10.1f us  %s\n' % (r[0], r[1][:45], r[3], r[2]))
print('ok')
