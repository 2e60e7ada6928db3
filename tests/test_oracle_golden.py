"""Pin the CPU oracle (oracle/ver_oracle.py) against vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only.  Tolerance: 1e-4 absolute in fp32
(BASELINE.json north_star), tighter where the arithmetic is short."""
import numpy as np
import pytest
import torch

import cases
from util import close, golden, maxdiff, oracle, pkg, state_from

T = torch.from_numpy
TOL = 1e-4


@pytest.mark.parametrize('name', list(cases.MSDA_CASES))
def test_msda_core_forward_backward(name):
    o = oracle()
    g = golden('msda_core_' + name)
    c = cases.msda_inputs(**cases.MSDA_CASES[name])
    value = T(c['value']).requires_grad_(True)
    loc = T(c['loc']).requires_grad_(True)
    w = T(c['w']).requires_grad_(True)
    out = o.msda_core(value, c['shapes'].tolist(), loc, w)
    out.backward(T(c['grad_out']))
    if name == 'vocc':
        assert maxdiff(out.detach()[:, ::5], g['out']) < 2e-5
        assert maxdiff(out.detach()[:, ::5], g['out_twin']) < 2e-5
        assert close(value.grad[:, ::3], g['grad_value'])
        assert close(loc.grad[:, ::5], g['grad_loc'])
        assert close(w.grad[:, ::5], g['grad_w'])
        norms = [float(t.double().norm()) for t in (out.detach(), value.grad, loc.grad, w.grad)]
        np.testing.assert_allclose(norms, g['norms'], rtol=1e-5)
    else:
        assert maxdiff(out.detach(), g['out']) < 2e-5
        assert maxdiff(out.detach(), g['out_twin']) < 2e-5
        assert maxdiff(value.grad, g['grad_value']) < 2e-5
        assert close(loc.grad, g['grad_loc'])
        assert maxdiff(w.grad, g['grad_w']) < 2e-5
    assert float(g['twin_maxdiff']) < 2e-5


@pytest.mark.parametrize('gname', list(cases.GRIDS))
def test_projection_and_visibility(gname):
    o = oracle()
    syn = pkg('synthetic')
    g = golden('point_sampling')
    z, h, w = cases.GRIDS[gname]
    ref3d = o.reference_points_3d(z, h, w)
    if gname != 'c2':
        assert maxdiff(ref3d, g[gname + '_ref3d']) == 0.0
    w2p, org = syn.camera_batch(2, seed=1)
    for b in range(2):
        uv, mask = o.point_sampling(ref3d, T(w2p[b]), T(org[b]), cases.PC_RANGE)
        key = '%s_b%d_' % (gname, b)
        want_mask = np.unpackbits(g[key + 'mask'], axis=1)[:, :z * h * w].astype(bool)
        assert np.array_equal(mask.numpy(), want_mask)          # bit-exact visibility
        assert mask.sum(1).tolist() == g[key + 'hits'].tolist()
        step = 16 if gname == 'c2' else 1
        assert maxdiff(uv[:, ::step], g[key + 'uv']) < 1e-5


def test_hits_match_survey_probe():
    g = golden('point_sampling')
    assert g['vocc_b0_hits'].tolist() == [115, 155, 155, 115, 155, 155]
    assert g['c2_b0_hits'].tolist() == [5142, 6423, 6423, 5142, 6423, 6423]


def test_msda3d_module():
    o = oracle()
    g = golden('msda3d_small')
    p = state_from(g)
    out = o.msda3d_forward(p, '', T(g['query']), T(g['value']), T(g['ref']), [(7, 7)], 4, 8)
    assert maxdiff(out, g['out']) < 2e-5


def test_sca_module():
    o = oracle()
    g = golden('sca_small')
    p = state_from(g)
    out = o.sca_forward(p, '', T(g['query']), T(g['feat'][:, :, 0]), T(g['uv'][:, 0]),
                        T(g['mask'][:, 0]), [(7, 7)], 4, 8)
    assert maxdiff(out, g['out']) < 2e-5


def test_encoder_small_forward_backward():
    o = oracle()
    syn = pkg('synthetic')
    g = golden('encoder_small')
    w2p, org = syn.camera_batch(2, seed=1)
    grid = tuple(int(v) for v in g['grid'])
    for b in range(2):
        p = {k: v.clone().requires_grad_(True) for k, v in state_from(g).items()}
        q = T(g['bev_query'][:, 0]).clone().requires_grad_(True)
        feat = T(g['feats'][b]).clone().requires_grad_(True)
        out = o.encoder_forward(p, '', q, feat, T(w2p[b]), T(org[b]), grid, cases.PC_RANGE, 2,
                                heads=4, points=8)
        assert maxdiff(out.detach(), g['out'][b]) < 2e-5
        gout = T(np.random.default_rng(40 + b).standard_normal(out.shape).astype(np.float32))
        out.backward(gout)
        assert close(q.grad, g['grad_query'][b][:, 0])
        assert close(feat.grad, g['grad_feats'][b][:, :, 0])
        for k, v in p.items():
            assert close(v.grad, g['gp%d.%s' % (b, k)], atol=2e-4, rtol=1e-4), k


@pytest.mark.parametrize('gname', ['vocc', 'c1'])
def test_get_voxel_features_full_width(gname):
    o = oracle()
    syn = pkg('synthetic')
    g = golden('encoder_vocc')
    z, h, w = cases.GRIDS[gname]
    shapes = {'cams_embeds': (6, 768), 'level_embeds': (4, 768)}
    p = _vocc_transformer_state(syn)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    bq = T(np.random.default_rng(5).standard_normal((z * h * w, 768)).astype(np.float32))
    for b in range(2):
        out = o.get_voxel_features(p, '', T(feats[b]).unsqueeze(1), bq, (z, h, w), T(w2p[b]),
                                   T(org[b]), cases.PC_RANGE)
        key = '%s_b%d_' % (gname, b)
        assert maxdiff(out[0, ::7], g[key + 'out']) < TOL
        assert abs(float(out.double().norm()) - float(g[key + 'norm'])) < 1e-2
    del shapes


def _vocc_transformer_state(syn):
    """State-dict of the vocc.py transformer (encoder only) from the seed-2 recipe: the key
    names + shapes are the reference's (SURVEY.md section 5 checkpoint row)."""
    shapes = [('cams_embeds', (6, 768)), ('level_embeds', (4, 768))]
    for lid in range(3):
        pre = 'encoder.layers.%d.' % lid
        att = pre + 'attentions.0.'
        shapes += [(att + 'deformable_attention.sampling_offsets.weight', (128, 768)),
                   (att + 'deformable_attention.sampling_offsets.bias', (128,)),
                   (att + 'deformable_attention.attention_weights.weight', (64, 768)),
                   (att + 'deformable_attention.attention_weights.bias', (64,)),
                   (att + 'deformable_attention.value_proj.weight', (768, 768)),
                   (att + 'deformable_attention.value_proj.bias', (768,)),
                   (att + 'output_proj.weight', (768, 768)), (att + 'output_proj.bias', (768,)),
                   (pre + 'ffns.0.layers.0.0.weight', (1536, 768)),
                   (pre + 'ffns.0.layers.0.0.bias', (1536,)),
                   (pre + 'ffns.0.layers.1.weight', (768, 1536)),
                   (pre + 'ffns.0.layers.1.bias', (768,)),
                   (pre + 'norms.0.weight', (768,)), (pre + 'norms.0.bias', (768,)),
                   (pre + 'norms.1.weight', (768,)), (pre + 'norms.1.bias', (768,))]
    return {k: T(v) for k, v in syn.seeded_state(shapes, 2).items()}


def test_positional_encoding():
    o = oracle()
    syn = pkg('synthetic')
    g = golden('encoder_vocc')
    sd = syn.seeded_state([('row_embed.weight', (15, 768)), ('col_embed.weight', (15, 768)),
                           ('z_embed.weight', (4, 768))], 6)
    pos = o.positional_encoding({k: T(v) for k, v in sd.items()}, '', 4, 15, 15)
    assert maxdiff(pos[0, ::16], g['pos_vocc']) < 1e-6


def test_conv_transpose_direct_matches_definition():
    """a10: the scatter-form restatement vs torch's ConvTranspose3d on a small case, and the
    even-lattice property (odd rows/cols of the output equal the bias exactly)."""
    o = oracle()
    rng = np.random.default_rng(3)
    x = T(rng.standard_normal((1, 5, 3, 4, 6)).astype(np.float32))
    wt = T(rng.standard_normal((5, 7, 3, 5, 5)).astype(np.float32))
    b = T(rng.standard_normal(7).astype(np.float32))
    y = o.conv_transpose3d_direct(x, wt, b, **o.UPSAMPLE_GEOM)
    ref = torch.nn.functional.conv_transpose3d(x, wt, b, **o.UPSAMPLE_GEOM)
    assert y.shape == ref.shape == (1, 7, 3, 8, 12)
    assert close(y, ref, atol=1e-5, rtol=1e-5)
    assert maxdiff(y[:, :, :, 1::2, :], b.view(1, 7, 1, 1, 1).expand(1, 7, 3, 4, 12)) == 0.0
    assert maxdiff(y[:, :, :, :, 1::2], b.view(1, 7, 1, 1, 1).expand(1, 7, 3, 8, 6)) == 0.0


def test_msda3d_init_pattern():
    o = oracle()
    bias = o.msda3d_init(8, 1, 8).view(8, 1, 8, 2)
    assert maxdiff(bias[0, 0, :, 0], torch.arange(1, 9).float()) < 1e-6   # head 0 -> +x ring
    assert float(bias.abs().max()) == pytest.approx(8.0, abs=1e-5)


@pytest.mark.parametrize('name', list(cases.MSDA3D_CASES))
def test_voxel_msda_core_forward_backward(name):
    """next-row 1: 3-D (trilinear) sampling op of the detection decoder vs the reference's own
    in-tree function (voxel_temporal_self_attention.py:275-335)."""
    o = oracle()
    g = golden('msda3d_core_' + name)
    c = cases.msda3d_inputs(**cases.MSDA3D_CASES[name])
    value = T(c['value']).requires_grad_(True)
    loc = T(c['loc']).requires_grad_(True)
    w = T(c['w']).requires_grad_(True)
    out = o.voxel_msda_core(value, c['shapes'].tolist(), loc, w)
    out.backward(T(c['grad_out']))
    assert maxdiff(out.detach(), g['out']) < 2e-5
    assert close(value.grad[:, ::3], g['grad_value'])
    assert close(loc.grad, g['grad_loc'])
    assert close(w.grad, g['grad_w'])


def test_lifting_path_oracle_matches_the_reference_head_vectors():
    """a10 pinned DIRECTLY: ``oracle.lifting_forward`` (encoder + ``occ_head_forward``, the reference's raw ``.view``s,
    ConvTranspose3d stack, occ_proj, occ_branches; head:554-580) on the seed-7 vocc.py head against the logits the
    reference's own ``VoxelFormerOccupancyHead.forward`` produced for the same parameters and inputs
    (tests/golden/head_vocc.npz, make_golden_head.py) -- no product code between the oracle and the reference's vectors
    (the package only names and seeds the parameters)."""
    o = oracle()
    syn = pkg('synthetic')
    g = golden('head_vocc')
    head = pkg('registry').build_head(cases.vocc_head_cfg())
    code_weights = head.code_weights.detach().clone()
    syn.load_seeded(head, 7)
    head.code_weights.data.copy_(code_weights)
    p = {k: v.detach().clone() for k, v in head.state_dict().items()}
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    with torch.no_grad():
        bev, occ = o.lifting_forward(p, T(feats[0]).unsqueeze(1), T(w2p[0]), T(org[0]))
    assert occ.shape == (1, 504000, 16)
    assert maxdiff(bev[0, ::7], g['c3_b0_bev']) < TOL
    assert close(occ[0, ::997], g['c3_b0_occ'], atol=1e-4, rtol=1e-4)
    assert abs(float(occ.double().norm()) - float(g['c3_b0_occ_norm'])) < 1e-4 * float(g['c3_b0_occ_norm'])
