import sys, os, numpy as np, torch, warnings
warnings.filterwarnings('ignore')
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.')
import cases
from util import golden, pkg
import test_head_gpu as th
T = torch.from_numpy
syn = pkg('synthetic'); g = golden('head_vocc')
w2p, org = syn.camera_batch(2, seed=1); feats = syn.vit_features(2, seed=0)
gg = T(np.random.default_rng(60).standard_normal((504000, 16)).astype(np.float32)).to('cuda')
head = th._head(cases.vocc_head_cfg(), 7)
with torch.autocast('cuda', dtype=torch.bfloat16):
    outs = head(T(feats[0]).to('cuda').unsqueeze(1), th._metas(w2p, org, [0]))
(outs['occupancy_preds'][0].float() * gg).sum().backward()
names = [str(s) for s in g['c3_grad_names']]
worst = []
for name, want in zip(names, g['c3_grad_norms']):
    got = float(dict(head.named_parameters())[name].grad.double().norm())
    worst.append((abs(got - want) / max(1e-3, abs(want)), name, got, float(want)))
worst.sort(reverse=True)
for w in worst[:6]: print('%.4f %s %.3f %.3f' % w)
