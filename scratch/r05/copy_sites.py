"""Where the dtype-cast copies of the encoder's [bs, 900, 768] activations come from: aten::copy_ / _to_copy events of one
step with their Python stacks (torch profiler, verbose experimental config).  Scratch tool."""
import importlib, os, sys, argparse, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from torch.profiler import profile, ProfilerActivity
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
for _ in range(2): step()
torch.cuda.synchronize()
cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True, experimental_config=cfg) as prof:
    step(); torch.cuda.synchronize()
want = (B, 900, 768)
seen = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::add_', 'aten::add') and e.input_shapes and tuple(e.input_shapes[0]) in (want, (B, 6, 196, 768)):
        st = [s for s in (e.stack or []) if 'vln-ver_amd' in s or 'bench.py' in s or 'autograd' in s][:4]
        seen[(e.name, str(e.input_shapes[0]), ' <- '.join(x.strip()[-90:] for x in st))] += 1
for k, v in seen.most_common(40):
    print(v, k)
