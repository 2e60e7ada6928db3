"""Host-side mirror of the reference's plugin API (no GPU): registry names, kwargs from the
vocc.py dicts, parameter names/shapes identical to what the reference built (keys recorded in
the golden fixtures), initialisation rules."""
import warnings

import numpy as np
import pytest
import torch

import cases
from util import golden, oracle, pkg

warnings.filterwarnings('ignore')


def test_registry_has_reference_type_names():
    r = pkg('registry')
    pkg()
    for name in ('SpatialCrossAttention', 'MSDeformableAttention3D'):
        assert name in r.ATTENTION.module_dict
    assert 'VoxelFormerEncoder' in r.TRANSFORMER_LAYER_SEQUENCE.module_dict
    assert 'VoxelFormerLayer' in r.TRANSFORMER_LAYER.module_dict
    assert 'MyCustomBaseTransformerLayer' in r.TRANSFORMER_LAYER.module_dict
    assert 'VoxelPerceptionTransformer' in r.TRANSFORMER.module_dict
    assert 'VoxelLearnedPositionalEncoding' in r.POSITIONAL_ENCODING.module_dict
    assert 'FFN' in r.FEEDFORWARD_NETWORK.module_dict


def test_build_from_cfg_errors_like_mmcv():
    r = pkg('registry')
    with pytest.raises(KeyError, match='not in the attention registry'):
        r.build_attention(dict(type='NoSuchAttention'))
    with pytest.raises(KeyError, match='must contain the key "type"'):
        r.build_attention(dict(embed_dims=3))
    with pytest.raises(ValueError, match='divisible'):
        r.build_attention(dict(type='MSDeformableAttention3D', embed_dims=30, num_heads=4))


def test_small_encoder_state_dict_matches_reference_keys():
    r = pkg('registry')
    pkg()
    g = golden('encoder_small')
    enc = r.build_transformer_layer_sequence(cases.small_encoder_cfg())
    ours = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    theirs = {k[3:]: g[k].shape for k in g.files if k.startswith('sd.')}
    assert ours == theirs
    enc.load_state_dict({k: torch.from_numpy(g['sd.' + k]) for k in theirs}, strict=True)


def test_vocc_transformer_builds_with_reference_names_and_sizes():
    r = pkg('registry')
    pkg()
    tr = r.build_transformer(cases.vocc_transformer_cfg())
    sd = tr.state_dict()
    assert sd['level_embeds'].shape == (4, 768) and sd['cams_embeds'].shape == (6, 768)
    pre = 'encoder.layers.2.attentions.0.'
    assert sd[pre + 'deformable_attention.sampling_offsets.weight'].shape == (128, 768)
    assert sd[pre + 'deformable_attention.attention_weights.weight'].shape == (64, 768)
    assert sd[pre + 'output_proj.weight'].shape == (768, 768)
    assert sd['encoder.layers.0.ffns.0.layers.0.0.weight'].shape == (1536, 768)
    assert sd['encoder.layers.0.ffns.0.layers.1.weight'].shape == (768, 1536)
    assert sum(v.numel() for v in sd.values()) == 11088192
    layer = tr.encoder.layers[0]
    assert layer.operation_order == ('cross_attn', 'norm', 'ffn', 'norm') and not layer.pre_norm
    assert layer.ffns[0].layers[0][2].p == 0.1            # ffn_dropout via the deprecated kwarg
    assert tr.encoder.num_points_in_voxel == 4 and tr.encoder.pc_range == list(cases.PC_RANGE)


def test_msda3d_init_matches_reference_rule():
    r = pkg('registry')
    pkg()
    att = r.build_attention(dict(type='MSDeformableAttention3D', embed_dims=768, num_points=8,
                                 num_levels=1))
    assert float(att.sampling_offsets.weight.abs().max()) == 0.0
    assert float(att.attention_weights.weight.abs().max()) == 0.0
    assert float(att.attention_weights.bias.abs().max()) == 0.0
    want = oracle().msda3d_init(8, 1, 8)
    assert torch.allclose(att.sampling_offsets.bias, want, atol=1e-6)
    assert float(att.value_proj.bias.abs().max()) == 0.0


def test_transformer_init_weights():
    r = pkg('registry')
    pkg()
    torch.manual_seed(0)
    tr = r.build_transformer(cases.vocc_transformer_cfg())
    tr.init_weights()
    att = tr.encoder.layers[1].attentions[0].deformable_attention
    assert float(att.sampling_offsets.weight.abs().max()) == 0.0       # re-init after xavier
    assert abs(float(tr.cams_embeds.std()) - 1.0) < 0.05
    w = tr.encoder.layers[0].ffns[0].layers[1].weight
    bound = (6.0 / (768 + 1536)) ** 0.5
    assert float(w.abs().max()) <= bound + 1e-6


def test_positional_encoding_matches_golden_on_cpu():
    r = pkg('registry')
    syn = pkg('synthetic')
    pkg()
    pe = r.build_positional_encoding(dict(type='VoxelLearnedPositionalEncoding', num_feats=384,
                                          row_num_embed=15, col_num_embed=15, z_num_embed=4))
    syn.load_seeded(pe, 6)
    pos = pe(torch.zeros(1, 4, 15, 15))
    assert pos.shape == (1, 768, 4, 15, 15)
    assert float((pos[0, ::16] - torch.from_numpy(golden('encoder_vocc')['pos_vocc'])).abs().max()) < 1e-6


def test_camera_store_reads_reference_formats(tmp_path):
    syn = pkg('synthetic')
    store_mod = pkg('camera_store')
    w2p, org = syn.camera_batch(2, seed=1)
    syn.write_camera_files(str(tmp_path), 'scanA', ['vp0', 'vp1'], w2p, org)
    store = store_mod.CameraStore(str(tmp_path))
    m, o = store.lookup('scanA_vp1')
    assert np.array_equal(m, w2p[1]) and np.array_equal(o, org[1])
    mb, ob = store.batch([[{'sample_idx': 'scanA_vp0'}], {'sample_idx': 'scanA_vp1'}])
    assert mb.shape == (2, 6, 4, 4) and np.array_equal(ob, org)
    with pytest.raises(KeyError):
        store.lookup('scanA_vp9')
    with pytest.raises(FileNotFoundError):
        store.lookup('scanB_vp0')


def test_get_reference_points_matches_oracle():
    enc_mod = pkg('modules.voxel_encoder')
    ref = enc_mod.VoxelFormerEncoder.get_reference_points(4, 15, 15, dim='3d', bs=2, device='cpu')
    assert ref.shape == (2, 1, 900, 3)
    assert torch.equal(ref[0, 0], oracle().reference_points_3d(4, 15, 15))
    assert torch.equal(ref[0, 0], torch.from_numpy(golden('point_sampling')['vocc_ref3d']))


def test_init_weights_chain_matches_reference_fixture():
    """a11 pinned: the reference's own ``init_weights`` chain on the vocc.py head (tests/golden/init_vocc.npz,
    generated by running the reference's files) against OUR head AND the oracle's restatement -- the same entries are
    re-initialised, constants are the same constants (zeroed projections, focal-prior biases -log 99), the offset
    ring is identical, and the random initialisers draw from the same distributions (std within 5 %, |max| inside the
    same xavier bound)."""
    from util import init_report
    g = golden('init_vocc')
    pkg()
    torch.manual_seed(0)
    head = pkg('registry').build_head(cases.vocc_head_cfg())
    rep = init_report(head)
    assert [str(n) for n in rep['names']] == [str(n) for n in g['rep_names']]
    names = [str(n) for n in g['rep_names']]
    bad = [n for n, a, b in zip(names, rep['changed'], g['rep_changed']) if bool(a) != bool(b)]
    assert not bad, 'init_weights touches a different set of entries: %s' % bad[:8]
    bad = [n for n, a, b in zip(names, rep['const'], g['rep_const']) if bool(a) != bool(b)]
    assert not bad, bad[:8]
    for n, c, a, b in zip(names, g['rep_const'], rep['value'], g['rep_value']):
        if c:
            assert a == pytest.approx(b, abs=1e-7), n
    for n, c, sa, sb, ma, mb in zip(names, g['rep_const'], rep['std'], g['rep_std'], rep['amax'], g['rep_amax']):
        if not c and 'sampling_offsets.bias' not in n:
            assert sa == pytest.approx(sb, rel=0.05), (n, sa, sb)
            if ma < 1.0:                          # bounded (uniform) initialisers: same bound
                assert ma == pytest.approx(mb, rel=0.02), (n, ma, mb)
    sd = head.state_dict()
    pre = 'transformer.encoder.layers.%d.attentions.0.deformable_attention.sampling_offsets.bias'
    for layer in range(3):
        assert torch.equal(sd[pre % layer], torch.from_numpy(g['enc_offset_bias']))
    assert torch.allclose(oracle().msda3d_init(8, 1, 8), torch.from_numpy(g['enc_offset_bias']), atol=1e-7)
    assert torch.equal(sd['transformer.decoder.layers.0.attentions.1.sampling_offsets.bias'],
                       torch.from_numpy(g['dec_offset_bias']))
    assert torch.allclose(sd['cls_branches.5.6.bias'], torch.from_numpy(g['cls_prior_bias']), atol=1e-7)
    assert torch.allclose(sd['occ_branches.6.bias'], torch.from_numpy(g['occ_prior_bias']), atol=1e-7)
    assert float(g['cls_prior_bias'][0]) == pytest.approx(-float(np.log(99.0)), abs=1e-6)


# ------------------------------------------------------------------------------- host logic added in round 5
def test_lent_parameters_swap_and_restore_on_cpu():
    """modules/lowp_params.py without a GPU: the bf16 copies are exact roundings of the masters laid out in one flat buffer,
    the owning modules hold them (and the bf16 Linear path) only inside the context, gradients come back in fp32 on the
    masters, a parameter the step does not use keeps ``grad is None``, and an exception inside the context still restores."""
    lp = pkg('modules.lowp_params')
    torch.manual_seed(4)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))
    mha = torch.nn.MultiheadAttention(16, 2)
    spare = torch.nn.Linear(16, 3)
    lent = lp.LowpParams([net, mha, spare])
    assert len(lent.slots) == 2 * 2 + 2 + 2 + 2 and all(o % 128 == 0 for o in lent.offsets)
    assert not lent.applies(torch.zeros(1))                                   # CPU tensors: autocast's own casts
    x = torch.randn(5, 3, 8)
    with lent.lent():
        assert net[0].weight.dtype == torch.bfloat16 and net[1].weight.dtype == torch.float32
        assert torch.equal(net[0].weight, net[0]._parameters['weight']) and 'forward' in net[0].__dict__
        assert 'forward' not in mha.out_proj.__dict__                        # MultiheadAttention calls F.linear itself
        assert torch.equal(net[0].weight.float(), net[0].weight.float().bfloat16().float())
        y = net[0](x)                                                        # the lent bf16 Linear
        assert y.dtype == torch.bfloat16 and tuple(y.shape) == (5, 3, 16)
        y.float().sum().backward()
    want = torch.nn.functional.linear(x.bfloat16(), net[0].weight.detach().bfloat16(), net[0].bias.detach().bfloat16())
    assert torch.equal(y.detach(), want)
    assert isinstance(net[0].weight, torch.nn.Parameter) and net[0].weight.dtype == torch.float32 and 'forward' not in net[0].__dict__
    assert net[0].weight.grad.dtype == torch.float32 and tuple(net[0].weight.grad.shape) == (16, 8)
    assert torch.allclose(net[0].bias.grad, torch.full((16,), 15.0))
    assert net[2].weight.grad is None and spare.weight.grad is None and mha.in_proj_weight.grad is None
    with pytest.raises(ZeroDivisionError):
        with lent.lent():
            1 / 0
    assert net[0].weight.dtype == torch.float32 and 'forward' not in net[0].__dict__


def test_box_denormalisation_over_whole_rows_equals_the_sliced_form():
    """``VoxelFormerOccupancyHead._denormalize`` (sigmoid + affine on columns 0, 1, 4 of whole rows) against the reference's
    slice-and-concatenate form (head:590-606), values and gradients, for fp32 and for bf16 branch outputs (promotion to the
    reference points' fp32 as in the reference)."""
    head = pkg('dense_heads.voxelformer_occupancy_head').VoxelFormerOccupancyHead
    torch.manual_seed(0)
    rng = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]

    def sliced(tmp, reference):
        xy = (tmp[..., 0:2] + reference[..., 0:2]).sigmoid()
        zc = (tmp[..., 4:5] + reference[..., 2:3]).sigmoid()
        x = xy[..., 0:1] * (rng[3] - rng[0]) + rng[0]
        y = xy[..., 1:2] * (rng[4] - rng[1]) + rng[1]
        z = zc * (rng[5] - rng[2]) + rng[2]
        return torch.cat([x, y, tmp[..., 2:4], z, tmp[..., 5:]], -1)
    tmp = torch.randn(6, 2, 5, 10, requires_grad=True)
    ref = torch.randn(6, 2, 5, 3, requires_grad=True)
    a, b = sliced(tmp, ref), head._denormalize(tmp, ref, rng)
    assert torch.allclose(a, b, atol=1e-6, rtol=1e-6)
    w = torch.randn_like(a)
    for ga, gb in zip(torch.autograd.grad((a * w).sum(), [tmp, ref]), torch.autograd.grad((b * w).sum(), [tmp, ref])):
        assert torch.allclose(ga, gb, atol=1e-6, rtol=1e-6)
    tb = tmp.detach().bfloat16()
    a, b = sliced(tb, ref.detach()), head._denormalize(tb, ref.detach(), rng)
    assert a.dtype == b.dtype == torch.float32 and torch.allclose(a, b, atol=1e-6, rtol=1e-6)


def test_occ_proj_run_structure_of_the_vocc_geometry_is_periodic():
    """The table ``ver_lattice_rows`` runs on (dense_heads/occ_proj_lattice.py::_periodic_row_map), derived on the host: for
    768 channels on the 4 x 60 x 60 lattice the raw view's runs tile the flat lattice with period 960 (204 | 204 | 192 | 180 |
    180), 2 880 rows per group; the element offsets of a 3-sample operand buffer follow the groups' order; a geometry without
    that structure has no table."""
    opl = pkg('dense_heads.occ_proj_lattice')
    plan = opl.get_plan(768, 4, 120, 120, 'cpu')
    rm = plan.row_map
    assert rm is not None and rm['period'] == 960 and rm['quarter'] == 768 * 60 * 60 and rm['n_rows'] == 2880
    assert rm['seg_off'] == [0, 204, 408, 600, 780] and rm['seg_len'] == [204, 204, 192, 180, 180]
    full, spans, total = opl._row_map_for(plan, 3)
    assert total == sum(3 * g.n_rows * g.k_aug for g in plan.groups) and [s for s, _ in spans] == sorted(s for s, _ in spans)
    for off, ln, gi, base, pitch, rows in zip(full['seg_off'], full['seg_len'], full['seg_group'], full['seg_base'], full['seg_pitch'], full['seg_rows']):
        g = plan.groups[gi]
        assert ln == g.run_len and pitch == g.k_aug and rows == g.n_rows and base == spans[gi][0] and base % 8 == 0
    small = opl.get_plan(16, 4, 40, 48, 'cpu')
    assert small is None or small.row_map is None


def test_clip_adamw_has_no_cpu_path():
    opt_mod = pkg('optim')
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    with pytest.raises(TypeError):
        opt_mod.ClipAdamW([p], max_norm=1.0).step()
    # several parameter groups are fine (vocc.py:260-267 paramwise_cfg), one clip norm for all of them
    two = opt_mod.ClipAdamW([dict(params=[torch.nn.Parameter(torch.ones(1))], lr=1e-3),
                             dict(params=[torch.nn.Parameter(torch.ones(1))], lr=1e-4, max_norm=2.0)], max_norm=1.0)
    with pytest.raises(ValueError, match='one max_norm'):
        two.step()
    assert float(opt_mod.ClipAdamW([torch.nn.Parameter(torch.ones(2))]).step()) == 0.0     # nothing has a gradient
