"""Where do the 96 copies of two viewpoints start to differ inside a 192-viewpoint bf16 forward?  (tests/test_regime_gpu.py)"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import cases
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
ups, opl = pkg('dense_heads.upsample'), pkg('dense_heads.occ_proj_lattice')


def pairs(t, bdim, name):
    t = t.float().movedim(bdim, 0)
    first = t[:2]
    n = float(first.norm())
    d = torch.stack([(t[b:b + 2] - first).norm() / n for b in range(2, B, 2)])
    nz = torch.stack([((t[b:b + 2] - first) != 0).float().mean() for b in range(2, B, 2)])
    print('%-28s rel L2 vs first pair: max %.2e mean %.2e (argmax pair %d); differing elements: max %.2e mean %.2e'
          % (name, float(d.max()), float(d.mean()), 1 + int(d.argmax()), float(nz.max()), float(nz.mean())), flush=True)


with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
    emb = head(feats, None, only_bev=True, world2pixel=w2p, origin=org)
    pairs(emb, 0, 'encoder output (fp32)')
    low = pkg('modules.bricks').lowp_view(emb)
    pairs(low, 0, 'encoder output (bf16 copy)')
    x = low.contiguous().view(B, 768, 4, 15, 15)
    convs = list(head.up_sample)
    dt = torch.bfloat16
    e = ups._channels_last(x, dt)
    bs = [m.bias.to(dt) for m in convs]
    e1 = ups._Layer0Z4.apply(e, None, bs[0], convs[0].weight)
    pairs(e1, 0, 'lattice 1 (z-split)')
    e2 = ups._LatticeLayerZ4.apply(e1, None, bs[1], bs[0], False, convs[1].weight)
    pairs(e2, 1, 'lattice 2 (planar)')
    e3 = ups._LatticeLayerZ4.apply(e2, None, bs[2], bs[1], True, convs[2].weight)
    pairs(e3, 1, 'lattice 3 (planar)')
    for own in (True, False):
        ups._OWN_GEMM = own
        e1 = ups._Layer0Z4.apply(e, None, bs[0], convs[0].weight)
        e2 = ups._LatticeLayerZ4.apply(e1, None, bs[1], bs[0], False, convs[1].weight)
        pairs(e2, 1, 'lattice 2, own gemm %d' % own)
    ups._OWN_GEMM = True
    occ = head.occupancy_from_volume(emb)
    pairs(occ, 0, 'logits')
