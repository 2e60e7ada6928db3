"""Per-layer divergence of the 96 copies inside the encoder (bf16 autocast and fp32)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import cases
import warnings; warnings.filterwarnings('ignore')
pkg = lambda s=None: importlib.import_module('vln-ver_amd' + ('.' + s if s else ''))
T = torch.from_numpy
DEV = 'cuda'
B = int(os.environ.get('B', 192))
head = pkg('registry').build_head(cases.vocc_head_cfg()).eval()
cw = head.code_weights.detach().clone(); pkg('synthetic').load_seeded(head, 7); head.code_weights.data.copy_(cw)
head = head.to(DEV)
syn = pkg('synthetic')
w2p, org = syn.camera_batch(2, seed=1)
feats = T(syn.vit_features(2, seed=0)).to(DEV).permute(1, 0, 2, 3).contiguous().repeat(1, B // 2, 1, 1).contiguous()
w2p, org = T(w2p).to(DEV).repeat(B // 2, 1, 1, 1).contiguous(), T(org).to(DEV).repeat(B // 2, 1).contiguous()
hip = pkg('hipops')
outs = {}
enc = head.transformer.encoder
for i, layer in enumerate(enc.layers):
    layer.register_forward_hook(lambda m, a, o, i=i: outs.__setitem__('layer%d' % i, o.detach().float().clone()))
real = hip.sca_gather
def spy(*a, **k):
    r = real(*a, **k)
    outs['slots%d' % sum(1 for x in outs if x.startswith('slots'))] = (r[0] if isinstance(r, tuple) else r).detach().float().clone()
    return r
hip.sca_gather = spy
sca_mod = pkg('modules.spatial_cross_attention')
if hasattr(sca_mod, 'sca_gather'):
    sca_mod.sca_gather = spy
hit = hip.project_points(w2p, org, head.point_cloud_range, 4, 15, 15)
cnt = hit.mask().sum(0)[:2, :, 0] if hit.mask().dim() == 4 else None


def pairs(t, name):
    first = t[:2]
    n = float(first.norm())
    d = torch.stack([(t[b:b + 2] - first).norm() / n for b in range(2, B, 2)])
    print('%-10s max %.2e mean %.2e median %.2e; top pairs: %s' % (name, float(d.max()), float(d.mean()), float(d.median()),
          ' '.join('%d:%.1e' % (2 + 2 * int(i), float(d[i])) for i in d.argsort(descending=True)[:6])), flush=True)
    return 2 + 2 * int(d.argmax())


for mode in ('bf16', 'fp32'):
    outs.clear()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=mode == 'bf16'):
        emb = head(feats, None, only_bev=True, world2pixel=w2p, origin=org)
    print('----', mode)
    for k in sorted(outs):
        b = pairs(outs[k], k)
        if k in ('slots0', 'layer0'):
            t = outs[k]
            rd = (t[b:b + 2] - t[:2]).abs().amax(-1)          # [2, Nq]
            top = rd.flatten().argsort(descending=True)[:8]
            print('    worst pair %d: rows (sample, voxel, |d|max, cams) %s' % (b, ' '.join(
                '(%d,%d,%.1e,%s)' % (int(i) // rd.shape[1], int(i) % rd.shape[1], float(rd.flatten()[i]),
                                     '-' if cnt is None else int(cnt[int(i) // rd.shape[1], int(i) % rd.shape[1]])) for i in top)),
                  'rows differing: %d of %d' % (int((rd > 0).sum()), rd.numel()))
