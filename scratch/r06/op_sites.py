"""Which aten ops launch the step's glue kernels, and from where: one eager lifting step at B viewpoints under the torch
profiler; leaf CPU ops that launched device kernels, grouped by (op, input shape, first frames inside the package).
    python3 scratch/r06/op_sites.py B [min_count]"""
import importlib, os, sys, argparse, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import warnings; warnings.filterwarnings('ignore')
import bench
from torch.profiler import profile, ProfilerActivity
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = argparse.Namespace(workload="vocc_c2f_train", dtype="bf16", micro=192, batch=B, config=None)
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.LiftTrainer(head, 192, 'bf16').to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt, update = bench.make_optimizer(params)
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
def step():
    loss = model(feats, w2p, org, gt); loss.backward(); update()
for _ in range(3): step()
torch.cuda.synchronize()
cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True, experimental_config=cfg) as prof:
    step(); torch.cuda.synchronize()
seen = collections.OrderedDict()
tot_k = tot_us = 0
for e in prof.events():
    if str(e.device_type).endswith('CPU') and e.kernels:
        if any(c.kernels for c in (e.cpu_children or [])):
            continue
        us = sum(k.duration for k in e.kernels)
        st = [s for s in (e.stack or []) if ('vln-ver_amd' in s or 'bench.py' in s)][:3]
        site = ' <- '.join(x.strip().split('vln-ver_amd/')[-1][-70:] for x in st) or ('(autograd) ' + ' / '.join(
            s.strip()[-50:] for s in (e.stack or [])[:2]))
        shp = str([tuple(s) for s in (e.input_shapes or [])[:3]])[:70]
        key = (e.name, shp, site)
        d = seen.setdefault(key, [0, 0.0, 0])
        d[0] += 1; d[1] += us; d[2] += len(e.kernels)
        tot_k += len(e.kernels); tot_us += us
print('leaf ops with kernels: %d kernels, %.1f us of kernel time' % (tot_k, tot_us))
for key, (c, us, nk) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    if key[0].startswith('aten::') or True:
        print('%3d ops %3d kernels %8.1f us  %-28s %-70s %s' % (c, nk, us, key[0][:28], key[1], key[2]))
