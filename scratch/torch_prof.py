"""Attribute device time of one training step to aten ops (with input shapes).  Scratch tool."""
import importlib, os, sys, argparse
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import bench
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--micro', type=int, default=8)
ap.add_argument('--out', default='gpurun_out/torch_prof.txt')
a = ap.parse_args()
args = argparse.Namespace(workload="vocc_c2f_train", dtype="bf16", micro=a.micro, batch=a.batch, config=None)
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.LiftTrainer(head, a.micro, 'bf16').to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
B = a.batch
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)

def step():
    loss = model(feats, w2p, org, gt); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)

for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
with open(a.out, 'w') as f:
    f.write(ka.table(sort_by='self_device_time_total', row_limit=400, max_name_column_width=60, max_shapes_column_width=110))
    f.write('\n\n==== kernels ====\n')
    f.write(prof.key_averages().table(sort_by='self_device_time_total', row_limit=60, max_name_column_width=110))
f = open(a.out.replace('.txt', '_stack.txt'), 'w')
f.write(prof.key_averages(group_by_stack_n=8).table(sort_by='self_device_time_total', row_limit=150, max_name_column_width=50, max_src_column_width=140))
f.close()
print('ok')
