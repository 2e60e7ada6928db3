"""Which aten ops launch the device-to-device memcpys / the largest plain copy kernels of a step?"""
import importlib, os, sys, argparse, collections
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
from torch.profiler import profile, ProfilerActivity
args = argparse.Namespace(workload='vocc_c2f_train', dtype='bf16', micro=192, batch=192)
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.LiftTrainer(head, 192, 'bf16').to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
B = 192
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
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA: continue
    for k in ev.kernels:
        if 'Memcpy' in k.name or 'direct_copy' in k.name:
            st = [s for s in (ev.stack or []) if 'vln-ver_amd' in s or 'bench.py' in s]
            key = (k.name[:30], ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else '?')
            agg[key][0] += k.duration; agg[key][1] += 1
for key, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print('%8.1f us %3d  %s' % (us, n, ' | '.join(key)))
