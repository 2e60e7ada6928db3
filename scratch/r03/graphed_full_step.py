"""Full multi-task step with the head's forward and backward replayed as two hipGraphs (torch.cuda.make_graphed_callables);
the Hungarian assignment, the loss terms, clip and AdamW stay eager.  Usage: graphed_full_step.py <B>"""
import importlib, os, sys, time, traceback, types
import numpy as np, torch
sys.path.insert(0, '.')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda', 0)
importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
args = types.SimpleNamespace(config=None, workload='vocc_full_train', dtype='bf16', micro=192)
pkg, syn, head, n_train = bench.build_model(args, dev)
head.train()
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
gts = [syn.detection_gt(seed=40 + i, num_gt=3 + i % 5) for i in range(B)]
gb = [torch.from_numpy(g[0][:, :7]).to(dev) for g in gts]; gl = [torch.from_numpy(g[1]).to(dev) for g in gts]

class Fwd(torch.nn.Module):
    def __init__(self, head):
        super().__init__(); self.head = head; self.plan = None
    def forward(self, feats, w2p, org):
        with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
            outs = self.head(feats, None, world2pixel=w2p, origin=org, occupancy_rows=True)
        occ, plan, bs = outs['occupancy_preds']
        self.plan = (plan, bs)
        return outs['all_cls_scores'].float(), outs['all_bbox_preds'].float(), occ
fwd = Fwd(head)
def losses_of(cls, box, occ):
    outs = dict(all_cls_scores=cls, all_bbox_preds=box, occupancy_preds=(occ,) + fwd.plan)
    return sum(head.loss(gb, gl, gt, outs).values())
# probing step: freeze what the step never touches
losses_of(*fwd(feats, w2p, org)).backward()
for p in head.parameters():
    if p.requires_grad and p.grad is None: p.requires_grad_(False)
    p.grad = None
params = [p for p in head.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
def make_step(f):
    def step():
        loss = losses_of(*f(feats, w2p, org))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)
        return loss
    return step
def timeit(fn, n=10):
    for _ in range(3): last = fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): last = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, float(last.detach())
if os.environ.get('LIKE_BENCH'):
    model = bench.FullTrainer(head, 'bf16').to(dev).train()
    if os.environ.get('NOCACHE'):
        class T2(torch.nn.Module):
            def __init__(self, head): super().__init__(); self.head = head
            def forward(self, feats, w2p, org, gt, gt_boxes, gt_labels):
                with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
                    outs = self.head(feats, None, world2pixel=w2p, origin=org, occupancy_rows=True)
                outs = {k: (v.float() if torch.is_tensor(v) and k != 'occupancy_preds' else v) for k, v in outs.items()}
                return sum(self.head.loss(gt_boxes, gt_labels, gt, outs).values())
        model = T2(head).to(dev).train()
    hip = importlib.import_module('vln-ver_amd.hipops')
    def bstep():
        loss = model(feats, w2p, org, gt, gb, gl); loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)
        return loss
    if os.environ['LIKE_BENCH'] == '3':
        bstep = make_step(fwd)
    for _ in range(2): bstep()
    torch.cuda.synchronize()
    timer = hip.KernelTimer()
    if os.environ['LIKE_BENCH'] == '2': hip.KERNEL_TIMER = timer
    for _ in range(2): last = bstep()
    torch.cuda.synchronize(); hip.KERNEL_TIMER = None
    print('bench-like eager steps done', float(last.detach()), len(timer.records), flush=True)
    if os.environ.get('DEL_LAST'): del last
ms, l = timeit(make_step(fwd)); print('eager:   B=%d %.1f ms per step (loss %.4f)' % (B, ms, l), flush=True)
import os
NB = int(os.environ.get('NB', B))
if NB != B:      # capture at a smaller batch than the eager steps ran at (what bench.py's latency records do)
    feats, w2p, org, gt, gb, gl = feats[:, :NB].contiguous(), w2p[:NB], org[:NB], gt[:NB], gb[:NB], gl[:NB]
    if os.environ.get('EAGER_FIRST'): make_step(fwd)()
try:
    if os.environ.get('USE_CLASS'):
        gh = importlib.import_module('vln-ver_amd.graphs').GraphedHead(head, feats, w2p, org)
        def graphed(f, w, o):
            outs = gh(f, w, o)
            fwd.plan = outs['occupancy_preds'][1:]
            return outs['all_cls_scores'], outs['all_bbox_preds'], outs['occupancy_preds'][0]
    else:
        graphed = torch.cuda.make_graphed_callables(fwd, (feats, w2p, org), allow_unused_input=True)
    ms, l = timeit(make_step(graphed)); print('graphed: B=%d %.1f ms per step (loss %.4f)' % (B, ms, l), flush=True)
except Exception:
    traceback.print_exc(); print('GRAPH FAILED')
