"""Head-level host logic on CPU: vocc.py dict builds, parameter names/shapes equal the
reference's (recorded from the reference module in tests/golden/head_vocc.npz), the even-lattice
upsample equals ConvTranspose3d, the occupancy branch equals the oracle's literal restatement."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from util import close, golden, maxdiff, oracle, pkg


@pytest.fixture(scope='module')
def head():
    pkg()
    syn = pkg('synthetic')
    h = pkg('registry').build_head(cases.vocc_head_cfg()).eval()
    syn.load_seeded(h, 7)
    return h


def test_state_dict_equals_reference(head):
    g = golden('head_vocc')
    ours = [(k, ','.join(str(d) for d in v.shape)) for k, v in head.state_dict().items()]
    theirs = list(zip([str(s) for s in g['sd_names']], [str(s) for s in g['sd_shapes']]))
    assert ours == theirs                      # 341 entries, same order, same shapes
    assert sum(p.numel() for p in head.parameters()) == 215991739


def test_restated_cfg_builds_variants():
    pkg()
    r = pkg('registry')
    h = r.build_head(cases.vocc_head_cfg(only_occ=True))
    assert h.transformer.decoder is None and not hasattr(h, 'cls_branches')
    assert h.loss_cls is None
    with pytest.raises(TypeError, match='loss_layout'):           # the reference builds loss_layout unconditionally
        r.build_head(dict(cases.vocc_head_cfg(), add_layout=True))


@pytest.mark.parametrize('Z', [4, 3])            # Z = 4: z-split path (two z taps, N = 2C); else the 27-tap path
def test_upsample_lattice_equals_conv_transpose(Z):
    up = pkg('dense_heads.upsample')
    torch.manual_seed(1)
    x = torch.randn(2, 6, Z, 5, 7, dtype=torch.float64, requires_grad=True)
    ws = [(torch.randn(6, 6, 3, 5, 5, dtype=torch.float64) * 0.1).requires_grad_(True) for _ in range(3)]
    bs = [torch.randn(6, dtype=torch.float64, requires_grad=True) for _ in range(3)]
    y = up.upsample_dense(x, ws, bs)
    r = x
    for w, b in zip(ws, bs):
        r = F.conv_transpose3d(r, w, b, **up.GEOM)
    assert y.shape == r.shape == (2, 6, Z, 40, 56)
    assert maxdiff(y, r) < 1e-12
    assert maxdiff(y[:, :, :, 1::2], bs[2].view(1, 6, 1, 1, 1).expand(2, 6, Z, 20, 56)) == 0.0
    g = torch.randn_like(r)
    for a, b in zip(torch.autograd.grad(y, [x] + ws + bs, g), torch.autograd.grad(r, [x] + ws + bs, g)):
        assert maxdiff(a, b) < 1e-10


@pytest.mark.parametrize('C,Z,Hl,Wl', [(8, 4, 3, 5), (16, 2, 4, 3), (24, 4, 5, 6)])
def test_occ_proj_on_lattice_equals_dense_view(C, Z, Hl, Wl):
    """occ_proj evaluated on the even lattice (gathered GEMMs, hand-written backward) vs the
    reference's raw .view + permute + Linear on the dense volume (head:564-571), values and all
    four gradients in fp64, bs = 3."""
    opl, ups = pkg('dense_heads.occ_proj_lattice'), pkg('dense_heads.upsample')
    gen = torch.Generator().manual_seed(C + Hl)
    bs, out = 3, 20
    e = torch.randn(bs, Z, Hl, Wl, C, generator=gen, dtype=torch.float64).requires_grad_(True)
    ub = torch.randn(C, generator=gen, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(out, Z * C, generator=gen, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(out, generator=gen, dtype=torch.float64).requires_grad_(True)
    res = opl.occ_proj_from_lattice(e, ub, w, b)
    assert res is not None
    rows, plan = res
    ours = opl.rows_to_voxels(rows, plan, bs)                                  # [bs, Hf*Wf, out]
    y = ups.full_volume(e, ub)                                                 # [bs, C, Z, 2Hl, 2Wl]
    dense = torch.nn.functional.linear(
        y.contiguous().view(bs, Z, 2 * Hl, 2 * Wl, C).permute(0, 2, 3, 1, 4).flatten(3), w, b)
    dense = dense.view(bs, -1, out)
    assert maxdiff(ours, dense) < 1e-11
    g = torch.randn(dense.shape, generator=gen, dtype=torch.float64)
    for a, r in zip(torch.autograd.grad(ours, [e, ub, w, b], g), torch.autograd.grad(dense, [e, ub, w, b], g)):
        assert maxdiff(a, r) < 1e-10


def test_occupancy_branch_equals_oracle(head):
    """a10 at full size on CPU: ours (lattice upsample) vs the oracle (ConvTranspose3d from its
    definition, raw views kept)."""
    o = oracle()
    emb = torch.from_numpy(np.random.default_rng(4).standard_normal((1, 900, 768)).astype(np.float32))
    with torch.no_grad():
        ours = head.occupancy_from_volume(emb)
        p = {k: v for k, v in head.state_dict().items()}
        want = o.occ_head_forward(p, '', emb[0], (4, 15, 15), (120, 120, 35), 128, refine_occ=True,
                                  use_torch_convt=False)
    assert ours.shape == want.shape == (1, 504000, 16)
    assert close(ours, want, atol=1e-4, rtol=1e-4)


def test_detection_and_occupancy_losses_match_reference():
    """next-row 2: our loss_single (Hungarian targets + focal / L1 / occupancy focal) vs the
    reference head's own loss_single (tests/golden/loss_vocc.npz)."""
    pkg()
    T = torch.from_numpy
    g = golden('loss_vocc')
    gh = golden('head_vocc')
    h = pkg('registry').build_head(dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG))
    cls = T(gh['c3_b0_cls'][-1]).clone().requires_grad_(True)
    box = T(gh['c3_b0_bbox'][-1]).clone().requires_grad_(True)
    boxes, labels = cases.detection_gt()
    logits, gt_occ = cases.occupancy_loss_inputs()
    occ = T(logits).requires_grad_(True)
    res = h.assigner.assign(box[0].detach(), cls[0].detach(), T(boxes), T(labels))
    assert res.gt_inds.tolist() == g['gt_inds'].tolist()                 # same Hungarian matching
    assert res.labels.tolist() == g['assigned_labels'].tolist()
    lc, lb, lo = h.loss_single(cls, box, occ, [T(boxes)], [T(labels)], T(gt_occ))
    assert float(lc) == pytest.approx(float(g['loss_cls']), rel=1e-5)
    assert float(lb) == pytest.approx(float(g['loss_bbox']), rel=1e-5)
    assert float(lo) == pytest.approx(float(g['loss_occ']), rel=1e-5)
    (lc + lb + lo).backward()
    assert close(cls.grad, g['grad_cls'], atol=1e-5, rtol=1e-4)
    assert close(box.grad, g['grad_box'], atol=1e-6, rtol=1e-4)
    assert close(occ.grad, g['grad_occ'], atol=1e-7, rtol=1e-4)
    # the dict of the reference's loss(): last layer + d0..d4
    preds = dict(all_cls_scores=T(gh['c3_b0_cls']), all_bbox_preds=T(gh['c3_b0_bbox']),
                 occupancy_preds=T(logits)[None])
    d = h.loss([T(boxes)[:, :7]], [labels], T(gt_occ)[None], preds)
    assert sorted(d) == sorted(['loss_cls', 'loss_bbox', 'loss_occupancy', 'loss_flow'] +
                               ['d%d.loss_%s' % (i, k) for i in range(5) for k in ('cls', 'bbox')])
    assert float(d['loss_cls']) == pytest.approx(float(g['loss_cls']), rel=1e-5)
    # loss() forms the Hungarian targets of all layers / samples in one batch: same numbers and
    # gradients as the reference-shaped per-layer loss_single, here on 2 samples with 3 and 0 boxes
    cls2 = T(np.concatenate([gh['c3_b0_cls'], gh['c3_b0_cls'][:, :, ::-1].copy()], 1)).requires_grad_(True)
    box2 = T(np.concatenate([gh['c3_b0_bbox'], gh['c3_b0_bbox'][:, :, ::-1].copy()], 1)).requires_grad_(True)
    gtb = [T(boxes)[:, :7], T(boxes)[:0, :7]]
    gtl = [T(labels), T(labels)[:0]]
    batched = h.loss(gtb, gtl, None, dict(all_cls_scores=cls2, all_bbox_preds=box2, occupancy_preds=None))
    gb = torch.autograd.grad(sum(v for k, v in batched.items() if 'cls' in k or 'bbox' in k), [cls2, box2])
    pad = [torch.cat([b, b.new_zeros(b.shape[0], 2)], 1) for b in gtb]
    tot = 0
    for lvl in range(cls2.shape[0]):
        lc_, lb_, _ = h.loss_single(cls2[lvl], box2[lvl], None, pad, gtl)
        key = '' if lvl == cls2.shape[0] - 1 else 'd%d.' % lvl
        assert float(batched[key + 'loss_cls']) == pytest.approx(float(lc_), rel=1e-6)
        assert float(batched[key + 'loss_bbox']) == pytest.approx(float(lb_), rel=1e-6)
        tot = tot + lc_ + lb_
    gs = torch.autograd.grad(tot, [cls2, box2])
    assert close(gb[0], gs[0], atol=1e-7, rtol=1e-5) and close(gb[1], gs[1], atol=1e-7, rtol=1e-5)
    # empty ground truth: everything is background, box loss is zero
    lc0, lb0, _ = h.loss_single(cls.detach(), box.detach(), None, [T(boxes)[:0]], [T(labels)[:0]])
    assert float(lb0) == 0.0 and float(lc0) > 0.0


def test_occupancy_postprocessing_and_metrics_match_reference(tmp_path):
    """next-rows 3/4: sparse occupancy prediction, SSC metrics, volume export layout."""
    pkg()
    T = torch.from_numpy
    g = golden('post_vocc')
    metrics = pkg('occupancy_metrics')
    vio = pkg('volume_io')
    h = pkg('registry').build_head(cases.vocc_head_cfg(only_occ=True))
    logits, gt = cases.occupancy_loss_inputs(seed=33, n=6000)
    res = h.get_occupancy_prediction(dict(occupancy_preds=T(logits)[None], flow_preds=None))
    assert np.array_equal(res['occupancy_preds'].numpy(), g['sparse'])
    dense = metrics.dense_labels(res['occupancy_preds'].numpy(), 6000, 16)
    m = metrics.SSCMetrics(17)
    m.add_batch(dense, gt)
    m.add_batch(dense[::-1].copy(), gt)
    st = m.get_stats()
    assert np.array_equal(m.hist, g['hist'])
    for k in ('iou', 'precision', 'recall', 'miou'):
        assert float(st[k]) == pytest.approx(float(g[k]), rel=1e-12)
    assert np.allclose(st['iou_ssc'], g['iou_ssc'], rtol=1e-12)
    # volume export: float64, raw (C,Z,H,W) reinterpretation of the [Nq,C] buffer, key = sample_idx
    emb = torch.arange(900 * 768, dtype=torch.float32).view(900, 768)
    w = vio.VolumeWriter(str(tmp_path / 'vols'))
    vol = h.export_volume(w, 'scanA_vp0', emb)
    back = vio.read_volume(str(tmp_path / 'vols'), 'scanA_vp0')
    assert back.dtype == np.float64 and back.shape == (768, 4, 15, 15) and np.array_equal(back, vol)
    assert back[1, 0, 0, 0] == 900.0 and back[0, 0, 0, 1] == 1.0       # flat[c*900 + k*225 + j*15 + i]
    # getbev=<path>: the forward itself appends the volumes of its viewpoints, keyed by sample_idx (head:627-638)
    h.getbev = str(tmp_path / 'dump')
    emb2 = torch.stack([emb, emb.flip(0)])
    h._dump_volumes(emb2, [dict(sample_idx='scanA_vp1'), dict(sample_idx='scanA_vp2')])
    assert np.array_equal(vio.read_volume(h.getbev, 'scanA_vp1'), vol)
    assert vio.read_volume(h.getbev, 'scanA_vp2')[0, 0, 0, 0] == float(emb[-1, 0])
    with pytest.raises(ValueError, match='sample_idx per viewpoint'):
        h._dump_volumes(emb2, None)
    h.getbev = None
    # feature store: CLS token dropped
    os_dir = tmp_path / 'feats'
    os_dir.mkdir()
    for d in range(6):
        np.save(str(os_dir / ('scanA_vp0_i1_%d.npy' % d)), np.full((1, 197, 768), d, dtype=np.float32))
    fs = vio.FeatureStore(str(os_dir))
    v = fs.viewpoint('scanA_vp0')
    assert v.shape == (6, 1, 196, 768) and float(v[3].mean()) == 3.0


def _layout_head(device='cpu'):
    pkg()
    h = pkg('registry').build_head(dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG, add_layout=True,
                                        loss_layout=dict(cases.LAYOUT_LOSS_CFG))).eval()
    cw = h.code_weights.detach().clone()
    pkg('synthetic').load_seeded(h, 7)
    h.code_weights.data.copy_(cw)
    return h.to(device)


def layout_loss_checks(h, g, outs_layout, device='cpu'):
    """Shared by the CPU and the GPU test: losses / gradients / dict / decoding of the layout-enabled head against
    the reference's (tests/golden/layout_vocc.npz), from given last-layer predictions."""
    T = torch.from_numpy
    gh = golden('head_vocc')
    boxes, labels = cases.detection_gt()
    lay = cases.layout_gt()
    nocc = 35 * 15 * 15
    gt_occ = np.full(nocc, 16, dtype=np.int64)
    gt_occ[g['occ_sparse'][:, 0]] = g['occ_sparse'][:, 1]
    cls = T(gh['c3_b0_cls'][-1]).to(device).requires_grad_(True)
    box = T(gh['c3_b0_bbox'][-1]).to(device).requires_grad_(True)
    lp = T(g['layout_preds'][-1]).to(device).requires_grad_(True)
    gtb = torch.cat([T(boxes)[:, :7], torch.zeros(len(boxes), 2)], 1).to(device)
    gtl = torch.cat([T(lay), torch.zeros(1, 2)], 1).to(device)
    res = h.assigner.assign(lp[0].detach(), None, gtl, torch.zeros(1, dtype=torch.long, device=device), None, layout=True)
    assert res.gt_inds.tolist() == g['layout_gt_inds'].tolist()
    lc, lb, ll, _ = h.loss_single_layout(cls, box, lp, None, [gtb], [T(labels).to(device)], [gtl])
    assert float(lc) == pytest.approx(float(g['loss_cls']), rel=1e-5)
    assert float(lb) == pytest.approx(float(g['loss_bbox']), rel=1e-5)
    assert float(ll) == pytest.approx(float(g['loss_layout']), rel=1e-5)
    (lc + lb + ll).backward()
    assert close(lp.grad.cpu(), g['grad_layout'], atol=1e-7, rtol=1e-4)
    assert close(cls.grad.cpu(), g['grad_cls'], atol=1e-5, rtol=1e-4)
    assert close(box.grad.cpu(), g['grad_box'], atol=1e-6, rtol=1e-4)
    # decoding (layout_coder.py) + get_layouts (bottom centre)
    dec = h.layout_coder.decode({'all_layout_preds': T(g['layout_preds']).to(device)})
    assert close(dec[0]['layouts'].cpu(), g['decoded'], atol=1e-5, rtol=1e-5)
    got = h.get_layouts({'all_layout_preds': T(g['layout_preds']).to(device)})[0][0].cpu()
    want = T(g['decoded']).clone()
    want[:, 2] -= want[:, 5] * 0.5
    assert close(got, want, atol=1e-5, rtol=1e-5)
    return gt_occ, gtb, gtl, labels


def test_layout_branch_losses_match_reference():
    """Room-layout branch (BASELINE configs[4] "room layout"; head:760-902, :992-1248): Hungarian matching on the
    layout L1 cost, the four loss terms with gradients, the loss dict, the layout coder."""
    T = torch.from_numpy
    g = golden('layout_vocc')
    gh = golden('head_vocc')
    h = _layout_head()
    gt_occ, gtb, gtl, labels = layout_loss_checks(h, g, None)
    # whole dict: stored cls / box / layout predictions of all six layers; the occupancy logits were stored strided,
    # so its term is checked through the occupancy_loss function on the stored rows instead
    preds = dict(all_cls_scores=T(gh['c3_b0_cls']), all_bbox_preds=T(gh['c3_b0_bbox']),
                 all_layout_preds=T(g['layout_preds']), occupancy_preds=None)
    d = h.loss_addlayout([gtb], [T(labels)], [gtl], None, preds)
    want = dict(zip([str(k) for k in g['dict_keys']], g['dict_vals']))
    assert sorted(d) == sorted(want)
    for k, v in want.items():
        if k != 'loss_occupancy':
            assert float(d[k]) == pytest.approx(v, rel=1e-5, abs=1e-7), k

    class Boxes:                                  # the reference's calling convention (bs = 1, mmdet3d box objects)
        def __init__(self, arr):
            self.tensor = T(arr)
            self.gravity_center = self.tensor[:, :3]
    boxes, _ = cases.detection_gt()
    d2 = h.loss_addlayout(Boxes(boxes[:, :7]), labels, Boxes(cases.layout_gt()), None, preds)
    for k in want:
        assert float(d2[k]) == float(d[k]), k
