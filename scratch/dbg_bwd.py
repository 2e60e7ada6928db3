import sys, os, importlib, warnings
warnings.filterwarnings('ignore')
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import cases
from util import golden, oracle, pkg, oracle_slots
T = torch.from_numpy
torch.backends.cuda.matmul.allow_tf32 = False
pkg(); r = pkg('registry'); syn = pkg('synthetic'); hip = pkg('hipops')
g = golden('encoder_vocc')
tr = r.build_transformer(cases.vocc_transformer_cfg()).eval(); syn.load_seeded(tr, 2); tr.to('cuda')
z,h,w = 4,15,15
w2p, org = syn.camera_batch(2, seed=1); feats = syn.vit_features(2, seed=0)
b = 0
bq = T(np.random.default_rng(5).standard_normal((900, 768)).astype(np.float32)).cuda().requires_grad_(True)
mlvl = T(feats[b]).cuda().unsqueeze(1).requires_grad_(True)
out = tr.get_voxel_features(mlvl, bq, z,h,w, bev_pos=None, world2pixel=T(w2p[b:b+1]).cuda(), origin=T(org[b:b+1]).cuda())
gout = T(np.random.default_rng(50 + b).standard_normal(out.shape).astype(np.float32)).cuda()
out.backward(gout)
got = mlvl.grad[:, 0, ::7].cpu(); want = T(g['vocc_b0_grad_feats'])
d = (got - want).abs()
print('per camera max err', d.amax((1,2)).tolist())
print('per key-row max err cam3', d[3].amax(1).tolist())
print('per head max err', d.view(6,28,8,96).amax((0,1,3)).tolist())
# oracle backward on CPU
o = oracle()
p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in tr.state_dict().items()}
mc = T(feats[b]).unsqueeze(1).requires_grad_(True); qc = bq.detach().cpu().clone().requires_grad_(True)
oc = o.get_voxel_features(p, '', mc, qc, (z,h,w), T(w2p[b]), T(org[b]), cases.PC_RANGE)
oc.backward(gout.cpu())
print('fwd hip-oracle', float((out.detach().cpu()-oc.detach()).abs().max()))
print('oracle vs golden grad_feats', float((mc.grad[:,0,::7]-want).abs().max()))
print('hip vs oracle grad_feats', float((mlvl.grad.cpu()-mc.grad).abs().max()))
print('hip vs oracle grad_query', float((bq.grad.cpu()-qc.grad).abs().max()))
for k, prm in tr.named_parameters():
    e = float((prm.grad.cpu() - p[k].grad).abs().max()); s = float(p[k].grad.abs().max())
    if e > 1e-3*max(s,1e-3): print('param', k, 'err', e, 'scale', s)
err = (bq.grad.cpu()-qc.grad).abs().amax(1)
top = err.topk(8)
print('top voxel errs', top.values.tolist(), top.indices.tolist())
hit = hip.project_points(T(w2p[b:b+1]).cuda(), T(org[b:b+1]).cuda(), cases.PC_RANGE, z,h,w)
vis = hit.vis[0].cpu()
print('vis bits of those', [int(vis[i]) for i in top.indices])
# recompute per-layer sample coordinates on CPU with oracle pieces
import torch.nn.functional as F
feat = mc.detach()[:,0] + p['cams_embeds'].detach()[:6,None,:] + p['level_embeds'].detach()[None,0:1,:]
x = qc.detach()[None]
ref3d = o.reference_points_3d(z,h,w); uv, mask = o.point_sampling(ref3d, T(w2p[b]), T(org[b]), cases.PC_RANGE)
pd = {k: v.detach() for k, v in p.items()}
for lid in range(3):
    pre = 'encoder.layers.%d.attentions.0.deformable_attention.' % lid
    off = F.linear(x[0], pd[pre+'sampling_offsets.weight'], pd[pre+'sampling_offsets.bias']).view(900,8,8,2)
    for vi in top.indices[:4].tolist():
        for c in range(6):
            if (int(vis[vi])>>c)&1:
                px = (uv[c,vi,0] + off[vi,:,:,0]/14)*14-0.5; py = (uv[c,vi,1]+off[vi,:,:,1]/14)*14-0.5
                dx = (px - px.round()).abs().min(); dy = (py-py.round()).abs().min()
                print('layer',lid,'voxel',vi,'cam',c,'min dist to integer x %.2e y %.2e' % (float(dx), float(dy)), 'uv', uv[c,vi].tolist())
    x = o.layer_forward(pd, 'encoder.layers.%d.' % lid, x, feat, uv[:,:,None,:], mask[:,:,None], [(14,14)], 8, 8)
