"""Find what goes non-finite in a replayed lifting step in train mode."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import warnings; warnings.filterwarnings('ignore')
import bench
sys.argv = sys.argv[:1]
args = bench.parse(); args.batch = args.micro = 1
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
pkg, syn, head, n_train = bench.build_model(args, dev)
mode = os.environ.get('MODE', 'train')
if mode == 'p0':
    for m in head.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
model = bench.LiftTrainer(head, 1, 'bf16').to(dev)
model.train(mode != 'eval')
params = [p for p in model.parameters() if p.requires_grad]
opt, update = bench.make_optimizer(params)
w2p_np, org_np = syn.camera_batch(1, seed=1)
feats = torch.from_numpy(syn.vit_features(1, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(1, head.voxel_num))).to(dev)
stash = []
orig_bwd = hip.SCAGatherFunction.backward
def spy_bwd(ctx, grad_slots):
    out = orig_bwd(ctx, grad_slots)
    value, offsets, logits = ctx.saved_tensors
    stash.append(dict(gs=grad_slots, gv=out[0], goff=out[1], glog=out[2], value=value, offsets=offsets, logits=logits, hit=ctx.hit))
    return out
hip.SCAGatherFunction.backward = staticmethod(spy_bwd)
rstash = []
orig_rd = hip.ReluDropoutFunction.backward
def spy_rd(ctx, grad_y):
    out = orig_rd(ctx, grad_y)
    rstash.append(dict(y=ctx.saved_tensors[0], gy=grad_y, gx=out[0]))
    return out
hip.ReluDropoutFunction.backward = staticmethod(spy_rd)
lift = importlib.import_module('vln-ver_amd.graphs').GraphedLiftStep(model, opt, feats, w2p, org, gt, warmup=1)
stash[:] = stash[-3:]                      # the capture's three backward calls (layer 2, 1, 0)
rstash[:] = rstash[-3:]
names = {id(p): k for k, p in model.named_parameters()}
for it in range(2):
    l = lift(*lift.inputs); torch.cuda.synchronize()
    badg = [names[id(p)] for p in params if p.grad is not None and not torch.isfinite(p.grad).all()]
    badp = [names[id(p)] for p in params if not torch.isfinite(p).all()]
    print('%s replay %d: loss %.6f norm %s; non-finite grads %d %s; params %d' % (mode, it, float(l), float(lift.grad_norm), len(badg), badg[:6], len(badp)), flush=True)
    for li, d in enumerate(stash if os.environ.get('STASH') else []):
        fin = {k: bool(torch.isfinite(v.float()).all()) for k, v in d.items() if torch.is_tensor(v)}
        h = d['hit']
        print('     sca backward call %d: finite %s; fwd_cnt %s vis_cnt %s' % (li, fin, h.fwd_cnt.flatten().tolist(), h.vis_cnt.flatten().tolist()), flush=True)
        if not fin['goff']:
            bad = (~torch.isfinite(d['goff'])).nonzero()
            print('       first non-finite d(offsets) entries:', bad[:6].tolist(), 'count', len(bad), 'vis of those voxels', d['hit'].vis[0, bad[:6, 1]].tolist())

import gc
bad = [(names[id(p)], p.grad) for p in params if p.grad is not None and not torch.isfinite(p.grad).all()]
allt = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda]
print('live cuda tensors', len(allt))
for name, g in bad[:4]:
    lo, hi = g.data_ptr(), g.data_ptr() + g.numel() * g.element_size()
    nanpos = (~torch.isfinite(g.flatten())).nonzero().flatten()
    print('NaN grad', name, tuple(g.shape), g.dtype, hex(lo), 'n_nonfinite', len(nanpos), 'first idx', nanpos[:8].tolist(), 'values', g.flatten()[nanpos[:4]].tolist(), 'finite sample', g.flatten()[:4].tolist(), 'is_view', g._is_view(), 'storage bytes', g.untyped_storage().nbytes())
    for o in allt:
        try:
            a, b = o.data_ptr(), o.data_ptr() + o.numel() * o.element_size()
        except Exception:
            continue
        if o is not g and a < hi and b > lo and o.numel():
            print('    overlaps:', tuple(o.shape), o.dtype, hex(a), 'base storage', hex(o.untyped_storage().data_ptr()), o.untyped_storage().nbytes())

for li, d in enumerate(rstash):
    y, gy, gx = d['y'].float(), d['gy'].float(), d['gx'].float()
    ref = gy * (y > 0)
    err = (gx - ref).abs()
    colbad = (err.reshape(-1, err.shape[-1]) > 1e-3 * (ref.abs().max() + 1e-30)).any(0).nonzero().flatten()
    print('relu-dropout backward call %d: gy finite %s absmax %.3e; gx vs gy*(y>0): max err %.3e, bad columns %d %s; gy dtype %s shape %s stride %s' % (
        li, bool(torch.isfinite(gy).all()), float(gy.abs().max()), float(err.max()), len(colbad), colbad[:10].tolist(), d['gy'].dtype, tuple(d['gy'].shape), d['gy'].stride()))
    big = (gy.abs() > 1e3).nonzero()
    print('     |gy| > 1e3 entries:', len(big), big[:6].tolist())
