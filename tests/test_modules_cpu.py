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
