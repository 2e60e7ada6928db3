import sys, os, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import cases
from test_encoder_gpu import _train_mode_encoder, _run_encoder, T, DEV
bricks = importlib.import_module('vln-ver_amd.modules.bricks')
syn = importlib.import_module('vln-ver_amd.synthetic')
grid = (2, 6, 5); nq = 60
rng = np.random.default_rng(5)
w2p_np, org_np = syn.camera_batch(2, seed=1)
w2p, org = T(w2p_np).to(DEV), T(org_np).to(DEV)
q = T(rng.standard_normal((nq, 2, 256)).astype(np.float32)).to(DEV)
feats = T(rng.standard_normal((6, 196, 2, 256)).astype(np.float32)).to(DEV)
enc = _train_mode_encoder().train()
g = T(rng.standard_normal((2, nq, 256)).astype(np.float32)).to(DEV)
v = T(rng.standard_normal(q.shape).astype(np.float32)).to(DEV)
for pset in (0.1, 0.0):
    for m in enc.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = pset
    for fused in (False, True):
        bricks._FUSED_ADD_LN = fused
        qg = q.clone().requires_grad_(True)
        torch.manual_seed(77)
        out = _run_encoder(enc, qg, feats, w2p, org, grid)
        (out * g).sum().backward()
        an = float((qg.grad.double() * v).sum())
        for eps in (2e-3, 1e-3, 5e-4, 2.5e-4):
            torch.manual_seed(77)
            fp = (_run_encoder(enc, q + eps * v, feats, w2p, org, grid).detach().double() * g).sum()
            torch.manual_seed(77)
            fm = (_run_encoder(enc, q - eps * v, feats, w2p, org, grid).detach().double() * g).sum()
            print('p=%.1f fused=%d eps=%.1e autograd %.5f fd %.5f' % (pset, fused, eps, an, float((fp - fm) / (2 * eps))))
