"""GEMM ledger of one training step (verdict r3 item 3): every hipBLASLt / rocBLAS kernel of the bench step attributed to the
aten op that launched it (with operand shapes), its FLOPs, time and TFLOP/s.
    python scratch/r04/gemm_ledger.py [--batch 192] [--micro 192] [--out gpurun_out/r04_gemm_ledger.csv]"""
import importlib, os, sys, argparse, json, collections, re
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=192)
ap.add_argument('--micro', type=int, default=192)
ap.add_argument('--workload', default='vocc_c2f_train')
ap.add_argument('--out', default='gpurun_out/r04_gemm_ledger.csv')
ap.add_argument('--no-tuned-gemms', action='store_true')
a = ap.parse_args()
args = argparse.Namespace(workload=a.workload, dtype='bf16', micro=a.micro, batch=a.batch, config=None)
dev = torch.device('cuda', 0)
hip = importlib.import_module('vln-ver_amd.hipops'); hip.lib()
if not a.no_tuned_gemms:
    importlib.import_module('vln-ver_amd.tuning').enable_tuned_gemms()
pkg, syn, head, n_train = bench.build_model(args, dev)
full = a.workload == 'vocc_full_train'
model = (bench.FullTrainer(head, 'bf16') if full else bench.LiftTrainer(head, a.micro, 'bf16')).to(dev).train()
B = a.batch
w2p_np, org_np = syn.camera_batch(B, seed=1)
feats = torch.from_numpy(syn.vit_features(B, seed=100)).to(dev).permute(1, 0, 2, 3).contiguous()
w2p, org = torch.from_numpy(w2p_np).to(dev), torch.from_numpy(org_np).to(dev)
gt = torch.from_numpy(np.random.default_rng(7).integers(0, 17, size=(B, head.voxel_num))).to(dev)
extra = ()
if full:
    gts = [syn.detection_gt(seed=40 + i, num_gt=3 + i % 5) for i in range(B)]
    extra = ([torch.from_numpy(g[0][:, :7]).to(dev) for g in gts], [torch.from_numpy(g[1]).to(dev) for g in gts])
    model(feats[:, :2], w2p[:2], org[:2], gt[:2], extra[0][:2], extra[1][:2]).backward()
    for p in model.parameters():
        if p.requires_grad and p.grad is None:
            p.requires_grad_(False)
        p.grad = None
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)

def step():
    loss = model(feats, w2p, org, gt, *extra); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 300.0); opt.step(); opt.zero_grad(set_to_none=True)

for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
trace = a.out.replace('.csv', '_trace.json')
prof.export_chrome_trace(trace)
ev = json.load(open(trace))['traceEvents']
os.remove(trace)
cpu_ops, kernels, launches = [], [], {}
for e in ev:
    if e.get('ph') != 'X':
        continue
    cat = e.get('cat', '')
    if cat == 'cpu_op':
        cpu_ops.append(e)
    elif cat == 'kernel':
        kernels.append(e)
    elif cat in ('cuda_runtime', 'cuda_driver') and 'correlation' in e.get('args', {}):
        launches[e['args']['correlation']] = e
cpu_ops.sort(key=lambda e: e['ts'])
GEMM_OPS = ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::linear', 'aten::matmul', 'aten::addmm_', 'aten::_scaled_mm')

import bisect
_starts = [op['ts'] for op in cpu_ops]


def owner(launch, any_op=False):
    """innermost (GEMM-like, or any) cpu op on the same thread whose interval contains the launch call"""
    best = None
    hi = bisect.bisect_right(_starts, launch['ts'])
    for op in cpu_ops[max(0, hi - 400):hi]:
        if op['tid'] == launch['tid'] and op['ts'] + op['dur'] >= launch['ts'] and (any_op or op['name'] in GEMM_OPS):
            best = op                       # later start = deeper nesting
    return best

def flops(name, shapes):
    try:
        if name in ('aten::mm',):
            (m, k), (k2, n) = shapes[0], shapes[1]; return 2.0 * m * n * k, (m, n, k, 1)
        if name in ('aten::addmm', 'aten::addmm_'):
            (m, k), (k2, n) = shapes[1], shapes[2]; return 2.0 * m * n * k, (m, n, k, 1)
        if name == 'aten::bmm':
            (b, m, k), (b2, k2, n) = shapes[0], shapes[1]; return 2.0 * b * m * n * k, (m, n, k, b)
        if name == 'aten::baddbmm':
            (b, m, k), (b2, k2, n) = shapes[1], shapes[2]; return 2.0 * b * m * n * k, (m, n, k, b)
    except Exception:
        pass
    return 0.0, None

rows = collections.OrderedDict()
other = collections.Counter()
other_ops = collections.Counter()
other_cnt = collections.Counter()
total_gpu = sum(k['dur'] for k in kernels)
for k in kernels:
    nm = k['name']
    is_gemm = nm.startswith(('Cijk', 'Custom_Cijk')) or 'gemm' in nm.lower()
    if not is_gemm:
        other[nm[:60]] += k['dur']
        la = launches.get(k['args'].get('correlation'))
        op = owner(la, True) if la else None
        other_ops[(op['name'] if op else '?', json.dumps(op['args'].get('Input Dims'))[:90] if op else '')] += k['dur']
        other_cnt[(op['name'] if op else '?', nm[:50])] += 1
        continue
    la = launches.get(k['args'].get('correlation'))
    op = owner(la) if la else None
    shapes = op['args'].get('Input Dims') if op else None
    strides = op['args'].get('Input Strides') if op else None
    key = (op['name'] if op else '?', json.dumps(shapes), json.dumps(strides), re.sub(r'_SN_.*', '', nm)[:90])
    r = rows.setdefault(key, dict(calls=0, us=0.0))
    r['calls'] += 1; r['us'] += k['dur']
with open(a.out, 'w') as f:
    f.write('# GEMM ledger of one %s step, B = %d viewpoints (torch profiler; TFLOP/s = useful flops / kernel time; MFMA peak 2500)\n' % (a.workload, B))
    f.write('op,M,N,K,batch,input_dims,input_strides,kernel,calls,total_ms,pct_of_gpu_time,TFLOPs_per_s,frac_of_2500\n')
    tot = 0.0
    for (op, shapes, strides, kn), r in sorted(rows.items(), key=lambda kv: -kv[1]['us']):
        fl, mnk = flops(op, json.loads(shapes) if shapes and shapes != 'null' else None)
        tf = fl * r['calls'] / (r['us'] * 1e-6) / 1e12 if r['us'] else 0.0
        tot += r['us']
        f.write('%s,%s,"%s","%s",%s,%d,%.3f,%.2f,%.0f,%.3f\n' % (op, ','.join(str(x) for x in (mnk or ('', '', '', ''))), shapes, strides, kn, r['calls'],
                                                         r['us'] / 1e3, 100.0 * r['us'] / total_gpu, tf, tf / 2500.0))
    f.write('# GEMM kernels total %.2f ms of %.2f ms GPU time (%.1f %%)\n' % (tot / 1e3, total_gpu / 1e3, 100.0 * tot / total_gpu))
    for nm, us in other.most_common(25):
        f.write('# other %-60s %.3f ms\n' % (nm, us / 1e3))
    for (opn, kn), n in other_cnt.most_common(60):
        f.write('# launches-by-op %-44s %-52s %d\n' % (opn, kn, n))
    for (nm, dims), us in other_ops.most_common(40):
        f.write('# other-by-op %-40s %-92s %.3f ms\n' % (nm, dims, us / 1e3))
print(open(a.out).read()[:6000])
