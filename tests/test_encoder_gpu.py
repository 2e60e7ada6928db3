"""Module-level parity on the GPU: SpatialCrossAttention, MSDeformableAttention3D,
VoxelFormerEncoder and VoxelPerceptionTransformer.get_voxel_features against the golden
vectors generated from the reference and against the CPU oracle.  ``-m gpu``."""
import warnings

import numpy as np
import pytest
import torch

import cases
from util import close, close_mostly, golden, maxdiff, oracle, pkg, state_from

warnings.filterwarnings('ignore')
pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda'
TOL = 1e-4


def _noTF32():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False


def test_msda3d_module_matches_reference():
    _noTF32()
    r = pkg('registry')
    pkg()
    g = golden('msda3d_small')
    att = r.build_attention(dict(type='MSDeformableAttention3D', embed_dims=32, num_heads=4,
                                 num_levels=1, num_points=8)).eval()
    att.load_state_dict(state_from(g), strict=True)
    att.to(DEV)
    out = att(T(g['query']).to(DEV), key=T(g['value']).to(DEV), value=T(g['value']).to(DEV),
              reference_points=T(g['ref']).to(DEV), spatial_shapes=torch.tensor([[7, 7]], device=DEV),
              level_start_index=torch.tensor([0], device=DEV))
    assert maxdiff(out.cpu(), g['out']) < 2e-5


def test_sca_module_matches_reference():
    _noTF32()
    r = pkg('registry')
    pkg()
    g = golden('sca_small')
    sca = r.build_attention(dict(
        type='SpatialCrossAttention', embed_dims=32, pc_range=list(cases.PC_RANGE),
        deformable_attention=dict(type='MSDeformableAttention3D', embed_dims=32, num_heads=4,
                                  num_levels=1, num_points=8))).eval()
    sca.load_state_dict(state_from(g), strict=True)
    sca.to(DEV)
    feat = T(g['feat']).to(DEV)
    out = sca(T(g['query']).to(DEV), feat, feat, reference_points_cam=T(g['uv']).to(DEV),
              bev_mask=T(g['mask']).to(DEV), spatial_shapes=torch.tensor([[7, 7]], device=DEV),
              level_start_index=torch.tensor([0], device=DEV))
    assert maxdiff(out.cpu(), g['out']) < 2e-5


def _metas(w2p, org, idx):
    return [{'sample_idx': 'scanA_vp%d' % b, 'world2pixel': w2p[b], 'origin': org[b]} for b in idx]


def test_encoder_small_forward_backward_matches_reference():
    _noTF32()
    r = pkg('registry')
    syn = pkg('synthetic')
    pkg()
    g = golden('encoder_small')
    enc = r.build_transformer_layer_sequence(cases.small_encoder_cfg()).eval()
    enc.load_state_dict(state_from(g), strict=True)
    enc.to(DEV)
    z, h, w = (int(v) for v in g['grid'])
    nq = z * h * w
    w2p, org = syn.camera_batch(2, seed=1)
    shapes = torch.tensor([[14, 14]], device=DEV)
    for b in range(2):
        enc.zero_grad()
        q = T(g['bev_query']).to(DEV).requires_grad_(True)                 # [Nq,1,C]
        feat = T(g['feats'][b]).to(DEV).unsqueeze(2).requires_grad_(True)  # [6,196,1,C]
        out = enc(q, feat, feat, bev_z=z, bev_h=h, bev_w=w, bev_pos=torch.zeros(nq, 1, 32, device=DEV),
                  spatial_shapes=shapes, level_start_index=torch.tensor([0], device=DEV),
                  prev_bev=None, shift=torch.zeros(1, 3, device=DEV), img_metas=_metas(w2p, org, [b]))
        assert maxdiff(out.detach().cpu(), g['out'][b]) < 2e-5
        gout = T(np.random.default_rng(40 + b).standard_normal(out.shape).astype(np.float32))
        out.backward(gout.to(DEV))
        assert close(q.grad.cpu(), g['grad_query'][b])
        assert close(feat.grad.cpu(), g['grad_feats'][b])
        for k, p in enc.named_parameters():
            assert close(p.grad.cpu(), g['gp%d.%s' % (b, k)], atol=2e-4, rtol=1e-4), k
    # both viewpoints in ONE batched call == the two bs=1 reference runs
    q2 = T(g['bev_query']).to(DEV).repeat(1, 2, 1)
    feat2 = T(g['feats']).to(DEV).permute(1, 2, 0, 3).contiguous()        # [6,196,2,C]
    out2 = enc(q2, feat2, feat2, bev_z=z, bev_h=h, bev_w=w, bev_pos=None, spatial_shapes=shapes,
               level_start_index=torch.tensor([0], device=DEV), prev_bev=None,
               img_metas=_metas(w2p, org, [0, 1]))
    assert maxdiff(out2.detach().cpu(), g['out'][:, 0]) < 2e-5


def _vocc_transformer():
    r = pkg('registry')
    syn = pkg('synthetic')
    pkg()
    tr = r.build_transformer(cases.vocc_transformer_cfg()).eval()
    syn.load_seeded(tr, 2)
    return tr.to(DEV)


@pytest.mark.parametrize('gname', ['vocc', 'c1', 'c2'])
def test_get_voxel_features_matches_reference(gname):
    """a1+a8 at full width (C=768), the three BASELINE.json grids; fp32, tolerance 1e-4."""
    _noTF32()
    syn = pkg('synthetic')
    g = golden('encoder_vocc')
    tr = _vocc_transformer()
    z, h, w = cases.GRIDS[gname]
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    bq = T(np.random.default_rng(5).standard_normal((z * h * w, 768)).astype(np.float32)).to(DEV)
    idx = [0] if gname == 'c2' else [0, 1]
    step = 97 if gname == 'c2' else 7
    with torch.no_grad():
        for b in idx:
            out = tr.get_voxel_features(T(feats[b]).to(DEV).unsqueeze(1), bq, z, h, w, bev_pos=None,
                                        img_metas=_metas(w2p, org, [b]))
            key = '%s_b%d_' % (gname, b)
            assert out.shape == (1, z * h * w, 768)
            assert maxdiff(out[0, ::step].cpu(), g[key + 'out']) < TOL
            assert abs(float(out.double().norm()) - float(g[key + 'norm'])) < 2e-2
        if gname != 'c2':     # batched call, device-resident camera tensors (the bench path)
            mlvl = T(feats).to(DEV).permute(1, 0, 2, 3).contiguous()
            out = tr.get_voxel_features(mlvl, bq, z, h, w, bev_pos=None,
                                        world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV))
            for b in idx:
                assert maxdiff(out[b, ::step].cpu(), g['%s_b%d_out' % (gname, b)]) < TOL


def test_get_voxel_features_backward_matches_reference():
    _noTF32()
    syn = pkg('synthetic')
    g = golden('encoder_vocc')
    tr = _vocc_transformer()
    z, h, w = cases.GRIDS['vocc']
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    for b in range(2):
        tr.zero_grad()
        bq = T(np.random.default_rng(5).standard_normal((z * h * w, 768)).astype(np.float32)).to(DEV)
        bq.requires_grad_(True)
        mlvl = T(feats[b]).to(DEV).unsqueeze(1).requires_grad_(True)
        out = tr.get_voxel_features(mlvl, bq, z, h, w, bev_pos=None, img_metas=_metas(w2p, org, [b]))
        gout = T(np.random.default_rng(50 + b).standard_normal(out.shape).astype(np.float32)).to(DEV)
        out.backward(gout)
        key = 'vocc_b%d_' % b
        assert close_mostly(bq.grad[::9].cpu(), g[key + 'grad_query'])
        assert close_mostly(mlvl.grad[:, 0, ::7].cpu(), g[key + 'grad_feats'])
        names = [str(s) for s in g[key + 'grad_names']]
        norms = g[key + 'grad_norms']
        ours = dict(query=float(bq.grad.double().norm()), feats=float(mlvl.grad.double().norm()))
        ours.update({k: float(p.grad.double().norm()) for k, p in tr.named_parameters()
                     if p.grad is not None})
        for name, want in zip(names, norms):
            assert abs(ours[name] - want) <= 5e-3 * max(1.0, abs(want)), name


def test_oracle_agrees_on_gpu_host():
    """The CPU oracle gives the golden answer on this host too (its BLAS may differ)."""
    from test_oracle_golden import test_get_voxel_features_full_width
    test_get_voxel_features_full_width('vocc')


def _train_mode_encoder(dims=256, seed=33):
    _noTF32()
    pkg()
    enc = pkg('registry').build_transformer_layer_sequence(
        cases.small_encoder_cfg(dims=dims, heads=8, points=8, ffn=2 * dims, layers=1))
    pkg('synthetic').load_seeded(enc, seed)
    return enc.to(DEV)


def _run_encoder(enc, q, feats, w2p, org, grid):
    z, h, w = grid
    return enc(q, feats, feats, bev_z=z, bev_h=h, bev_w=w, bev_pos=None,
               spatial_shapes=torch.tensor([[14, 14]], device=DEV), level_start_index=torch.tensor([0], device=DEV),
               prev_bev=None, world2pixel=w2p, origin=org)


def test_encoder_layer_training_mode_dropout_fused_vs_plain():
    """Training mode (dropout p = 0.1 in the attention tail, the FFN hidden block and the FFN tail): the fused passes
    (``ver_add_ln_*``, ``ver_relu_dropout_*``: keep-mask = hash(seed, index)) against the plain torch sequence
    (``VER_FUSED_ADD_LN=0``).  The masks differ by construction, so the comparison is
    (1) statistical -- the mean output over 48 mask draws agrees with the plain path's as well as two independent
        plain-path means agree with each other;
    (2) the gradient THROUGH the kept mask -- with the generator re-seeded the fused step is a deterministic function,
        and its autograd gradient matches a central finite difference along a random direction;
    (3) a Dropout module switched to eval() inside a train()-mode parent does not drop (the unfused semantics)."""
    bricks = pkg('modules.bricks')
    syn = pkg('synthetic')
    grid = (2, 6, 5)
    nq = 60
    rng = np.random.default_rng(5)
    w2p_np, org_np = syn.camera_batch(2, seed=1)
    w2p, org = T(w2p_np).to(DEV), T(org_np).to(DEV)
    q = T(rng.standard_normal((nq, 2, 256)).astype(np.float32)).to(DEV)
    feats = T(rng.standard_normal((6, 196, 2, 256)).astype(np.float32)).to(DEV)
    enc = _train_mode_encoder().train()

    def mean_out(fused, reps, seed0):
        old = bricks._FUSED_ADD_LN
        bricks._FUSED_ADD_LN = fused
        try:
            acc = 0
            with torch.no_grad():
                for r in range(reps):
                    torch.manual_seed(seed0 + r)
                    acc = acc + _run_encoder(enc, q, feats, w2p, org, grid).double()
            return acc / reps
        finally:
            bricks._FUSED_ADD_LN = old

    reps = 48
    plain_a, plain_b = mean_out(False, reps, 1000), mean_out(False, reps, 2000)
    fused = mean_out(True, reps, 3000)
    noise = float((plain_a - plain_b).norm() / plain_a.norm())
    diff = float((fused - plain_a).norm() / plain_a.norm())
    print('dropout statistics: plain-vs-plain %.4f fused-vs-plain %.4f' % (noise, diff))
    assert noise > 1e-3                    # dropout is really on
    assert diff < 1.3 * noise
    # (2) gradient through the kept mask: with the generator re-seeded the step is a deterministic function; its autograd
    #     gradient against central finite differences along 8 random directions.  (A finite difference of this layer
    #     carries ~1.5 % noise of its own from bilinear / ReLU kinks -- the same with dropout off, where the gradients
    #     are pinned by the golden vectors -- hence a vector comparison; a backward pass that ignored the mask would be
    #     off by ~30 %.  The exact op-level checks are test_add_dropout_layer_norm_fused / test_relu_dropout_fused.)
    g = T(rng.standard_normal((2, nq, 256)).astype(np.float32)).to(DEV)
    eps = 5e-4
    for fused_path in (False, True):
        old = bricks._FUSED_ADD_LN
        bricks._FUSED_ADD_LN = fused_path
        try:
            qg = q.clone().requires_grad_(True)
            torch.manual_seed(77)
            out = _run_encoder(enc, qg, feats, w2p, org, grid)
            torch.manual_seed(77)
            assert torch.equal(out.detach(), _run_encoder(enc, q, feats, w2p, org, grid).detach())   # same seed, same masks
            (out * g).sum().backward()
            an, fd = [], []
            for k in range(8):
                v = T(np.random.default_rng(100 + k).standard_normal(q.shape).astype(np.float32)).to(DEV)
                torch.manual_seed(77)
                fp = (_run_encoder(enc, q + eps * v, feats, w2p, org, grid).detach().double() * g).sum()
                torch.manual_seed(77)
                fm = (_run_encoder(enc, q - eps * v, feats, w2p, org, grid).detach().double() * g).sum()
                fd.append(float((fp - fm) / (2 * eps)))
                an.append(float((qg.grad.double() * v).sum()))
        finally:
            bricks._FUSED_ADD_LN = old
        an, fd = np.array(an), np.array(fd)
        rel = float(np.linalg.norm(fd - an) / np.linalg.norm(an))
        print('directional derivatives (%s): rel. L2 of finite differences vs autograd %.4f' % ('fused' if fused_path else 'plain', rel))
        assert rel < 5e-2, (an, fd)
    # (3) only the Dropout modules in eval(): nothing is dropped although the parent is in train()
    for m in enc.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    with torch.no_grad():
        a = _run_encoder(enc, q, feats, w2p, org, grid)
        enc.eval()
        b = _run_encoder(enc, q, feats, w2p, org, grid)
    assert torch.equal(a, b)


def test_tall_linear_weight_gradients_equal_the_plain_linear():
    """GEMM ledger (profiles/r04_gemm_ledger.csv): the weight gradients of the encoder's Linears over ~2e5 rows ran at
    0.15-0.6 PFLOP/s as single GEMMs with 9-18 output tiles; ``bricks.tall_linear`` routes them through the split-row
    batched form (dense_heads/row_linear.py).  Same outputs, same gradients: the vocc.py encoder at 20 viewpoints
    (18 000 voxel rows, 23 520 token rows: above the threshold) against the same encoder with the routing disabled."""
    from util import rel_l2
    bricks = pkg('modules.bricks')
    syn = pkg('synthetic')
    B = 20
    head = pkg('registry').build_head(cases.vocc_head_cfg())
    syn.load_seeded(head, 7)
    enc = head.transformer.encoder.to(DEV).train()
    for m in enc.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    w2p, org = syn.camera_batch(B, seed=3)
    gen = torch.Generator(device='cpu').manual_seed(1)
    q = torch.randn(900, B, 768, generator=gen).to(DEV)
    feats = torch.randn(6, 196, B, 768, generator=gen).to(DEV)
    gout = torch.randn(B, 900, 768, generator=gen).to(DEV)
    res = {}
    old = bricks._TALL_ROWS
    try:
        for name, rows in (('tall', 16000), ('plain', 10 ** 12)):
            bricks._TALL_ROWS = rows
            for p in enc.parameters():
                p.grad = None
            qq = q.clone().requires_grad_(True)
            out = enc(qq, feats, feats, bev_z=4, bev_h=15, bev_w=15, bev_pos=None,
                      spatial_shapes=torch.tensor([[14, 14]], device=DEV), level_start_index=torch.tensor([0], device=DEV),
                      prev_bev=None, world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV))
            out.backward(gout)
            res[name] = (out.detach().cpu(), qq.grad.cpu(),
                         {k: p.grad.detach().cpu() for k, p in enc.named_parameters() if p.grad is not None})
    finally:
        bricks._TALL_ROWS = old
    assert float((res['tall'][0] - res['plain'][0]).abs().max()) <= 1e-5
    assert rel_l2(res['tall'][1], res['plain'][1]) < 1e-5
    assert set(res['tall'][2]) == set(res['plain'][2]) and len(res['tall'][2]) >= 30
    for k, g in res['plain'][2].items():
        assert rel_l2(res['tall'][2][k], g) < 2e-5, k
