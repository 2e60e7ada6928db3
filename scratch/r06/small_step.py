"""The lifting step at a small batch, eager or as one hipGraph (graphs.GraphedLiftStep), for kernel traces:
    python3 scratch/r06/small_step.py B steps graph(0|1)
Prints ms per step; under rocprofv3 --kernel-trace the last `steps` steps are a clean window (marker kernels: none needed,
the summary script below takes the last steps * launches_per_step dispatches)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import warnings; warnings.filterwarnings('ignore')
import bench
B, steps, graph = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
args = bench.parse.__wrapped__() if hasattr(bench.parse, '__wrapped__') else None
sys.argv = sys.argv[:1]
args = bench.parse()
args.batch, args.micro = B, B
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
if os.environ.get('PYTORCH_TUNABLEOP_TUNING') != '1':
    importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
pkg, syn, head, n_train = bench.build_model(args, dev)
model = bench.LiftTrainer(head, B, 'bf16').to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt, update = bench.make_optimizer(params)
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
if graph:
    lift = importlib.import_module('vln-ver_amd.graphs').GraphedLiftStep(model, opt, feats, w2p, org, gt)
    f, w, o, g = lift.inputs
    step = lambda: lift(f, w, o, g)
else:
    def step():
        loss = model(feats, w2p, org, gt)
        loss.backward()
        update()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print('B=%d graph=%d: %.3f ms per step' % (B, graph, (time.perf_counter() - t0) / steps * 1e3), flush=True)
bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
print('non-finite parameters after the run:', len(bad), bad[:5], flush=True)
if graph:
    for i in range(3):
        l = step(); torch.cuda.synchronize()
        print('  replay loss %.6f grad norm %s' % (float(l), float(lift.grad_norm)), flush=True)
