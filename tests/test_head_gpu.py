"""Head-level parity on the GPU against vectors from the reference's own
``VoxelFormerOccupancyHead`` built from vocc.py (tests/golden/make_golden_head.py).
fp32, tolerance 1e-4 (north_star) on the occupancy logits.  ``-m gpu``."""
import contextlib
import warnings

import numpy as np
import pytest
import torch

import cases
from util import close, golden, maxdiff, pkg, rel_l2

warnings.filterwarnings('ignore')
pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = 'cuda'


def _head(cfg, seed):
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    pkg()
    h = pkg('registry').build_head(cfg).eval()
    code_weights = h.code_weights.detach().clone()
    pkg('synthetic').load_seeded(h, seed)
    h.code_weights.data.copy_(code_weights)      # (the seeded fill also hits this fixed, non-trainable loss-weight vector)
    return h.to(DEV)


def _metas(w2p, org, idx):
    return [{'sample_idx': 'scanA_vp%d' % b, 'world2pixel': w2p[b], 'origin': org[b]} for b in idx]


def test_vocc_head_forward_matches_reference():
    """C3 shape: vocc.py as written, 15x15x4 -> 120x120x35x16 (+ det branches)."""
    syn = pkg('synthetic')
    g = golden('head_vocc')
    head = _head(cases.vocc_head_cfg(), 7)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    with torch.no_grad():
        for b in range(2):
            outs = head(T(feats[b]).to(DEV).unsqueeze(1), _metas(w2p, org, [b]))
            key = 'c3_b%d_' % b
            occ = outs['occupancy_preds']
            assert occ.shape == (1, 504000, 16)
            assert maxdiff(outs['bev_embed'][::7, 0].cpu(), g[key + 'bev']) < 1e-4
            assert close(occ[0, ::997].cpu(), g[key + 'occ'], atol=1e-4, rtol=1e-4)
            assert abs(float(occ.double().norm()) - float(g[key + 'occ_norm'])) < 1e-4 * float(g[key + 'occ_norm'])
            assert close(outs['all_cls_scores'].cpu(), g[key + 'cls'], atol=2e-4, rtol=1e-4)
            assert close(outs['all_bbox_preds'].cpu(), g[key + 'bbox'], atol=2e-4, rtol=1e-4)
        # two viewpoints in one batched call (device-resident cameras)
        mlvl = T(feats).to(DEV).permute(1, 0, 2, 3).contiguous()
        outs = head(mlvl, None, world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV))
        for b in range(2):
            assert close(outs['occupancy_preds'][b, ::997].cpu(), g['c3_b%d_occ' % b], atol=1e-4, rtol=1e-4)
            assert close(outs['all_bbox_preds'][:, b].cpu(), g['c3_b%d_bbox' % b][:, 0], atol=2e-4, rtol=1e-4)


def test_vocc_head_occupancy_backward_matches_reference():
    syn = pkg('synthetic')
    g = golden('head_vocc')
    head = _head(cases.vocc_head_cfg(), 7)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    outs = head(T(feats[0]).to(DEV).unsqueeze(1), _metas(w2p, org, [0]))
    gg = T(np.random.default_rng(60).standard_normal((504000, 16)).astype(np.float32)).to(DEV)
    (outs['occupancy_preds'][0] * gg).sum().backward()
    names = [str(s) for s in g['c3_grad_names']]
    norms = g['c3_grad_norms']
    ours = {k: float(p.grad.double().norm()) for k, p in head.named_parameters() if p.grad is not None}
    assert set(names) <= set(ours)
    for name, want in zip(names, norms):
        assert abs(ours[name] - want) <= 1e-2 * max(1e-3, abs(want)), (name, ours[name], want)
    # gradients of a sum over 8e6 logits through 57600-term fp32 dot products (and a few ReLU
    # kinks, see util.close_mostly): relative L2 instead of an element-wise bound
    assert rel_l2(head.voxel_embedding.weight.grad[::9].cpu(), g['c3_grad_voxel_embedding']) < 5e-3
    assert rel_l2(head.up_sample[0].weight.grad[::37, ::41].cpu(), g['c3_grad_up0']) < 5e-3
    assert rel_l2(head.occ_proj.weight.grad[::53, ::29].cpu(), g['c3_grad_occ_proj']) < 5e-3


def test_single_scale_head_forward_matches_reference():
    """C2 shape: 50x50x16 single-scale volume (refine_occ=False), forward only."""
    syn = pkg('synthetic')
    g = golden('head_vocc')
    head = _head(cases.vocc_head_cfg(bev=(16, 50, 50), refine_occ=False), 8)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    with torch.no_grad():
        outs = head(T(feats[0]).to(DEV).unsqueeze(1), _metas(w2p, org, [0]))
    occ = outs['occupancy_preds']
    assert occ.shape == (1, 50 * 50 * 35, 16)
    assert maxdiff(outs['bev_embed'][::97, 0].cpu(), g['c2_b0_bev']) < 1e-4
    assert close(occ[0, ::173].cpu(), g['c2_b0_occ'], atol=1e-4, rtol=1e-4)
    assert abs(float(occ.double().norm()) - float(g['c2_b0_occ_norm'])) < 1e-4 * float(g['c2_b0_occ_norm'])


def test_lift_equals_default_branch():
    syn = pkg('synthetic')
    head = _head(cases.vocc_head_cfg(), 7)
    w2p, org = syn.camera_batch(1, seed=1)
    feats = syn.vit_features(1, seed=0)
    mlvl = T(feats[0]).to(DEV).unsqueeze(1)
    with torch.no_grad():
        emb, occ = head.lift(mlvl, _metas(w2p, org, [0]))
        outs = head(mlvl, _metas(w2p, org, [0]))
    assert torch.equal(emb[0], outs['bev_embed'][:, 0])
    assert torch.equal(occ, outs['occupancy_preds'])


# Largest single-element deviation of the bf16 head from the reference's fp32 logits, as MEASURED on MI355X (round 4, this
# test with -s: see the printed line) plus a 1.5x margin; the north star's "1e-2 bf16" is held as relative L2 next to it.
MAX_ABS_BF16_HEAD = 2.5e-2     # measured: max |d| 1.49e-2 (rel L2 5.2e-3, mean |d| 2.8e-3) on logits of rms 0.67, max 2.77


def test_vocc_head_bf16_autocast_within_1e2():
    """BASELINE.json north_star: 1e-2 in bf16.  The bench's arithmetic (bf16 autocast GEMMs and
    lattices, bf16 value tile, fp32 gather / LayerNorm statistics / loss) against the reference's
    fp32 vectors: occupancy logits within 1e-2 relative L2 and 3e-2 absolute on |logit| ~ 5."""
    syn = pkg('synthetic')
    g = golden('head_vocc')
    head = _head(cases.vocc_head_cfg(), 7)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        emb, occ = head.lift(T(feats[0]).to(DEV).unsqueeze(1), _metas(w2p, org, [0]))
    occ = occ.float()
    want = T(g['c3_b0_occ'])
    got = occ[0, ::997].cpu()
    rl2, mx, mean = rel_l2(got, want), maxdiff(got, want), float((got - want).abs().mean())
    print('bf16 head vs fp32 reference: rel L2 %.3e, max |d| %.3e, mean |d| %.3e on logits with max |x| %.2f, rms %.2f'
          % (rl2, mx, mean, float(want.abs().max()), float(want.pow(2).mean().sqrt())))
    assert rl2 < 1e-2
    assert mx < MAX_ABS_BF16_HEAD
    assert mean < 2e-2
    assert abs(float(occ.double().norm()) - float(g['c3_b0_occ_norm'])) < 1e-2 * float(g['c3_b0_occ_norm'])
    assert rel_l2(emb[0, ::7].float().cpu(), T(g['c3_b0_bev'])) < 1e-2


def test_vocc_head_bf16_backward_with_folded_first_linear():
    """bf16 autocast backward of the bench's arithmetic.  On this path ``occ_branches[0]`` is folded into
    ``occ_proj`` (two Linears in a row) and its weight gradient comes back through the folded product:
    gradient norms of every parameter against the reference's fp32 vectors within bf16 accuracy, and the
    gradients of the layers around the fold against the UNFOLDED bf16 path (same kernels, reference order) --
    no further from our fp32 path than that one is."""
    syn = pkg('synthetic')
    g = golden('head_vocc')
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    gg = T(np.random.default_rng(60).standard_normal((504000, 16)).astype(np.float32)).to(DEV)
    grads = {}
    for mode in ('fp32', 'bf16_unfolded', 'bf16'):
        head = _head(cases.vocc_head_cfg(), 7)
        head.fold_first_occ_linear = mode == 'bf16'
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=mode != 'fp32'):
            outs = head(T(feats[0]).to(DEV).unsqueeze(1), _metas(w2p, org, [0]))
        (outs['occupancy_preds'][0].float() * gg).sum().backward()
        grads[mode] = {k: p.grad.float().cpu() for k, p in head.named_parameters() if p.grad is not None}
    names = [str(s) for s in g['c3_grad_names']]
    assert set(names) <= set(grads['bf16'])
    for name, want in zip(names, g['c3_grad_norms']):
        got = float(grads['bf16'][name].double().norm())
        # (bf16 arithmetic: the worst parameter, layers.2 attention_weights.bias, is 3.0 % off with plain torch ops in
        #  the encoder and 3.1 % with the fused residual + LayerNorm; everything else is within 1.7 %)
        assert abs(got - want) <= 5e-2 * max(1e-3, abs(want)), (name, got, want)
    for name in ('occ_proj.weight', 'occ_proj.bias', 'occ_branches.0.weight', 'occ_branches.0.bias',
                 'occ_branches.1.weight', 'occ_branches.3.weight', 'up_sample.2.weight'):
        folded = rel_l2(grads['bf16'][name], grads['fp32'][name])
        unfolded = rel_l2(grads['bf16_unfolded'][name], grads['fp32'][name])
        print(name, 'folded', folded, 'unfolded', unfolded)
        assert folded < max(1.25 * unfolded, 1e-2), (name, folded, unfolded)


@pytest.mark.parametrize('autocast', [False, True])
def test_occupancy_loss_in_row_order_equals_voxel_order(autocast):
    """``occupancy_loss_from_volume`` (logits left in the GEMMs' row order, targets permuted to match) against
    ``occupancy_loss(occupancy_from_volume(...))``: the same (logit row, target) pairs, so the same loss (up to the
    order of an fp32 sum over 8 M terms) and the same gradients."""
    syn = pkg('synthetic')
    w2p, org = syn.camera_batch(2, seed=1)
    feats = T(syn.vit_features(2, seed=0)).to(DEV).permute(1, 0, 2, 3).contiguous()
    gt = T(np.random.default_rng(5).integers(0, 17, size=(2, 504000))).to(DEV)
    res = {}
    for route in ('voxels', 'rows'):
        head = _head(cases.vocc_head_cfg(), 7).train()
        for m in head.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            emb = head(feats, None, only_bev=True, world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV))
            if route == 'rows':
                loss = head.occupancy_loss_from_volume(emb, gt)
            else:
                loss = head.occupancy_loss(head.occupancy_from_volume(emb), gt)
        loss.backward()
        res[route] = (float(loss), {k: p.grad.float().cpu() for k, p in head.named_parameters() if p.grad is not None})
    assert abs(res['rows'][0] - res['voxels'][0]) <= 1e-5 * abs(res['voxels'][0]), (res['rows'][0], res['voxels'][0])
    assert set(res['rows'][1]) == set(res['voxels'][1])
    # bf16: the row-order route is the fused MLP + focal-loss Function (round 4), whose d(logits) is rounded to bf16 BEFORE the
    # scalar factor of the loss is applied and once more after it; the voxel-order route rounds once.  The two routes then differ
    # by the rounding noise of bf16 logit gradients themselves (measured 6e-3 on the most cancellation-prone parameter,
    # transformer.level_embeds; the bf16 path as a whole sits 7 % from the fp32 one, test_vocc_head_bf16_backward_...)
    for k, g in res['voxels'][1].items():
        assert rel_l2(res['rows'][1][k], g) < (1.5e-2 if autocast else 1e-5), k


def test_head_occupancy_loss_is_loud_about_a_bad_label_on_a_later_step():
    """``occupancy_loss`` wraps the focal term in ``nan_to_num`` (as the reference does), so the
    kernel's NaN for an out-of-range label would become a silent zero loss with zero gradients from the second step on
    (the host-side range check only runs on a module's first fused call).  The kernel's sticky device flag and its
    asynchronous host mirror keep the failure loud at the head level: the step after the bad one raises."""
    hip = pkg('hipops')
    losses = pkg('dense_heads.losses')
    head = _head(cases.vocc_head_cfg(), 7)
    gen = torch.Generator(device='cpu').manual_seed(11)
    logits = torch.randn(1, 8192, 16, generator=gen).to(DEV).requires_grad_(True)
    good = torch.randint(0, 17, (1, 8192), generator=gen).to(DEV)
    bad = good.clone()
    bad[0, 4321] = 23
    flag = hip.LabelRangeFlag.of(logits.device)
    flag.reset()
    try:
        assert float(head.occupancy_loss(logits, good)) > 0.0               # step 1: first fused call, host check passes
        silent = head.occupancy_loss(logits, bad)                            # step 2: NaN cleaned to 0 by nan_to_num ...
        assert float(silent) == 0.0
        with pytest.raises(RuntimeError, match='outside'):                   # ... but the next step is loud
            torch.cuda.synchronize()
            head.occupancy_loss(logits, good)
        with pytest.raises(RuntimeError, match='outside'):
            losses.FocalLoss.check_labels()
    finally:
        flag.reset()


def test_occupancy_postprocessing_on_gpu_is_bit_exact():
    """SURVEY 8f row 4 on the device path: ``get_occupancy_prediction`` (head:1505-1540) on GPU logits runs
    ``ver_occ_predict`` and returns the SAME (index, class) pairs as the reference produced on the CPU
    (tests/golden/post_vocc.npz), bit for bit; bf16 logits, ties, the threshold boundary, NaN and sizes that do not
    fill a block are held to the reference formulation evaluated by torch on the CPU."""
    g = golden('post_vocc')
    h = pkg('registry').build_head(cases.vocc_head_cfg(only_occ=True))
    logits, _ = cases.occupancy_loss_inputs(seed=33, n=6000)
    res = h.get_occupancy_prediction(dict(occupancy_preds=T(logits)[None].to(DEV), flow_preds=None))
    assert res['occupancy_preds'].is_cuda and res['occupancy_preds'].dtype == torch.int64
    assert np.array_equal(res['occupancy_preds'].cpu().numpy(), g['sparse'])
    assert res['flow_preds'] is None

    def reference(x, thr=0.25):                       # the reference's statements, on the CPU in fp32
        p = x.float().sigmoid()
        p = torch.cat((p, torch.ones_like(p)[:, :1] * thr), dim=-1)
        cls = p.argmax(dim=-1)
        idx, = torch.where(cls < x.shape[1])
        return torch.stack([idx, cls[idx]], dim=-1)

    gen = torch.Generator().manual_seed(5)
    for n in (1, 255, 1024, 1025, 70001):
        x = torch.randn(n, 16, generator=gen) * 2 - 1.5
        x[::7] = x[::7, :1]                            # ties: every class equal -> the first one wins
        x[3 % n] = float('nan')                        # NaN is the maximum for torch.argmax
        x[5 % n, 4:] = -30.0
        for xx in (x, x.bfloat16()):
            got = h.get_occupancy_prediction(dict(occupancy_preds=xx.to(DEV)))['occupancy_preds'].cpu()
            assert torch.equal(got, reference(xx)), (n, xx.dtype)
    empty = h.get_occupancy_prediction(dict(occupancy_preds=torch.full((300, 16), -9.0, device=DEV)))['occupancy_preds']
    assert empty.shape == (0, 2)
    none = h.get_occupancy_prediction(dict(occupancy_preds=torch.zeros(0, 16, device=DEV)))['occupancy_preds']
    assert none.shape == (0, 2)


def test_loss_single_on_gpu_matches_reference():
    """BASELINE configs[4], loss side, on the device: the reference head's own ``loss_single`` vectors
    (tests/golden/loss_vocc.npz; head:1251-1384) reproduced by OUR head living on the GPU -- Hungarian matching
    identical, the three loss values to 1e-5, their gradients as on the CPU."""
    g = golden('loss_vocc')
    gh = golden('head_vocc')
    h = _head(dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG), 7)
    cls = T(gh['c3_b0_cls'][-1]).to(DEV).requires_grad_(True)
    box = T(gh['c3_b0_bbox'][-1]).to(DEV).requires_grad_(True)
    boxes, labels = cases.detection_gt()
    logits, gt_occ = cases.occupancy_loss_inputs()
    occ = T(logits).to(DEV).requires_grad_(True)
    gb, gl = T(boxes).to(DEV), T(labels).to(DEV)
    res = h.assigner.assign(box[0].detach(), cls[0].detach(), gb, gl)
    assert res.gt_inds.tolist() == g['gt_inds'].tolist()
    assert res.labels.tolist() == g['assigned_labels'].tolist()
    lc, lb, lo = h.loss_single(cls, box, occ, [gb], [gl], T(gt_occ).to(DEV))
    assert float(lc) == pytest.approx(float(g['loss_cls']), rel=1e-5)
    assert float(lb) == pytest.approx(float(g['loss_bbox']), rel=1e-5)
    assert float(lo) == pytest.approx(float(g['loss_occ']), rel=1e-5)
    (lc + lb + lo).backward()
    assert close(cls.grad.cpu(), g['grad_cls'], atol=1e-5, rtol=1e-4)
    assert close(box.grad.cpu(), g['grad_box'], atol=1e-6, rtol=1e-4)
    assert close(occ.grad.cpu(), g['grad_occ'], atol=1e-7, rtol=1e-4)


def test_full_multitask_head_forward_and_loss_on_gpu():
    """BASELINE configs[4] end to end on one GPU in fp32: ``head(...)`` + ``head.loss(...)`` (head:903-990) for the two
    golden viewpoints.  The head outputs are pinned to the reference (test_vocc_head_forward_matches_reference); here
    the loss dict computed on the GPU from the GPU outputs must equal the loss dict our CPU implementation (pinned by
    loss_vocc.npz) computes from the REFERENCE's outputs, and the occupancy term must equal the oracle's focal loss of
    the reference logits' own statistics path (fused kernel, 1 M rows)."""
    syn = pkg('synthetic')
    gh = golden('head_vocc')
    cfg = dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG)
    head = _head(cfg, 7)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = T(syn.vit_features(2, seed=0)).to(DEV).permute(1, 0, 2, 3).contiguous()
    gts = [cases.detection_gt(seed=40 + i, num_gt=3 + i) for i in range(2)]
    gt_occ = T(np.random.default_rng(9).integers(0, 17, size=(2, 504000)))
    outs = head(feats, None, world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV))
    losses = head.loss([T(b[:, :7]).to(DEV) for b, _ in gts], [T(l).to(DEV) for _, l in gts], gt_occ.to(DEV), outs)
    keys = sorted(['loss_cls', 'loss_bbox', 'loss_occupancy', 'loss_flow'] +
                  ['d%d.loss_%s' % (i, k) for i in range(5) for k in ('cls', 'bbox')])
    assert sorted(losses) == keys
    # the same dict on the CPU from the reference's own head outputs
    pkg()
    cpu = pkg('registry').build_head(cfg).eval()
    ref_outs = dict(all_cls_scores=T(np.concatenate([gh['c3_b0_cls'], gh['c3_b1_cls']], 1)),
                    all_bbox_preds=T(np.concatenate([gh['c3_b0_bbox'], gh['c3_b1_bbox']], 1)), occupancy_preds=None)
    want = cpu.loss([T(b[:, :7]) for b, _ in gts], [T(l) for _, l in gts], None, ref_outs)
    for k in keys:
        if 'cls' in k or 'bbox' in k:
            assert float(losses[k]) == pytest.approx(float(want[k]), rel=1e-3, abs=1e-6), k
    from util import oracle
    occ_want = oracle().focal_loss(outs['occupancy_preds'].detach().cpu().reshape(-1, 16), gt_occ.reshape(-1),
                                   avg_factor=float((gt_occ < 16).sum()))
    assert float(losses['loss_occupancy']) == pytest.approx(float(occ_want), rel=2e-5)
    total = sum(losses.values())
    total.backward()
    assert torch.isfinite(total)
    grads = {k: p.grad for k, p in head.named_parameters() if p.grad is not None}
    for k in ('cls_branches.5.6.weight', 'reg_branches.0.4.weight', 'transformer.decoder.layers.0.attentions.1.value_proj.weight',
              'transformer.encoder.layers.0.attentions.0.deformable_attention.sampling_offsets.weight', 'occ_proj.weight',
              'up_sample.0.weight', 'query_embedding.weight', 'voxel_embedding.weight'):
        assert k in grads and torch.isfinite(grads[k]).all() and float(grads[k].abs().max()) > 0, k


def test_hungarian_targets_started_in_forward_give_the_same_losses():
    """``head(..., targets_for=(gt_boxes, gt_labels))`` (round 5): the cls / reg branches run before the occupancy head and
    the cost matrices of the Hungarian assignment (head:642-705) leave for the host right behind them, so that the host
    solves them while the GPU is busy with the occupancy head; ``loss`` picks the result up.  Same outputs as the plain
    forward (the branch order does not matter), the SAME loss dict to the last bit (identical cost matrices, identical
    assignment), and ``loss`` falls back to computing the targets itself when it is handed other ground truth."""
    syn = pkg('synthetic')
    cfg = dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG)
    head = _head(cfg, 7)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = T(syn.vit_features(2, seed=0)).to(DEV).permute(1, 0, 2, 3).contiguous()
    gts = [cases.detection_gt(seed=40 + i, num_gt=3 + i) for i in range(2)]
    gb, gl = [T(b[:, :7]).to(DEV) for b, _ in gts], [T(l).to(DEV) for _, l in gts]
    gt_occ = T(np.random.default_rng(9).integers(0, 17, size=(2, 504000))).to(DEV)
    with torch.no_grad():
        plain = head(feats, None, world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV))
        early = head(feats, None, world2pixel=T(w2p).to(DEV), origin=T(org).to(DEV), targets_for=(gb, gl))
        assert 'pending_targets' not in plain and early['pending_targets']['event'] is not None
        for k in ('all_cls_scores', 'all_bbox_preds', 'occupancy_preds', 'bev_embed'):
            assert torch.equal(plain[k], early[k]), k
        want = head.loss(gb, gl, gt_occ, plain)
        got = head.loss(gb, gl, gt_occ, early)
        assert sorted(want) == sorted(got)
        for k in want:
            assert float(want[k]) == float(got[k]), k
        # other ground truth than the one the targets were started for: recomputed, not reused
        other = [g.clone() for g in gb]
        other[0][:, 0] += 1.0
        redo = head.loss(other, gl, gt_occ, early)
        ref = head.loss(other, gl, gt_occ, plain)
        assert float(redo['loss_bbox']) == float(ref['loss_bbox']) != float(want['loss_bbox'])


def test_grouped_bf16_parameter_copies_equal_autocast(monkeypatch):
    """Under bf16 autocast a training step lends the decoder and the cls / reg branches bf16 copies of their Linear
    parameters made by ONE multi-tensor cast, and runs their plain Linears through a bf16 Function whose weight gradient
    is ``ver_wgrad_tn`` (modules/lowp_params.py; VER_LOWP_PARAMS=0: autocast's cast per parameter and call, the library's
    GEMMs).  The copies hold the values autocast would have produced; the Linear adds its bias inside the GEMM where
    autocast's 3-D path rounds to bf16 first, so outputs and losses agree to bf16 rounding, not to the bit.  Gradients
    arrive in fp32 on the fp32 masters; parameters the step does not touch keep ``grad is None`` and every module has its
    fp32 parameters (and its class's forward) back afterwards."""
    syn = pkg('synthetic')
    cfg = dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG)
    head = _head(cfg, 7)
    w2p, org = syn.camera_batch(2, seed=1)
    w2p, org = T(w2p).to(DEV), T(org).to(DEV)
    feats = T(syn.vit_features(2, seed=0)).to(DEV).permute(1, 0, 2, 3).contiguous()
    gts = [cases.detection_gt(seed=40 + i, num_gt=3 + i) for i in range(2)]
    gb, gl = [T(b[:, :7]).to(DEV) for b, _ in gts], [T(l).to(DEV) for _, l in gts]
    gt_occ = T(np.random.default_rng(9).integers(0, 17, size=(2, 504000))).to(DEV)
    results = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('VER_LOWP_PARAMS', mode)
        head.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            outs = head(feats, None, world2pixel=w2p, origin=org, targets_for=(gb, gl))
        outs = {k: (v.float() if torch.is_tensor(v) else v) for k, v in outs.items()}
        losses = head.loss(gb, gl, gt_occ, outs)
        # (gradients of a FIXED functional of the outputs: the Hungarian assignment may legitimately flip between two
        #  bf16 evaluations of an untrained head, and with it whole rows of the box gradients)
        probe = sum((outs[k] * torch.sin(torch.arange(outs[k].numel(), device=DEV).view_as(outs[k]) * 0.37)).sum()
                    for k in ('all_cls_scores', 'all_bbox_preds'))
        probe.backward()
        results[mode] = ({k: outs[k].detach().clone() for k in ('all_cls_scores', 'all_bbox_preds')},
                         {k: float(v) for k, v in losses.items()},
                         {k: p.grad.detach().clone() for k, p in head.named_parameters() if p.grad is not None})
    for k, p in head.named_parameters():
        assert isinstance(p, torch.nn.Parameter) and p.dtype == torch.float32 and p.is_leaf, k
    (o0, l0, g0), (o1, l1, g1) = results['0'], results['1']
    for m in head.modules():
        assert 'forward' not in m.__dict__
    for k in o0:
        assert rel_l2(o1[k].cpu().numpy(), o0[k].cpu().numpy()) < 3e-2, k      # (two bf16 evaluations of 6 decoder layers)
    # (the loss dict itself is not compared term by term: between two bf16 evaluations of an untrained head the Hungarian
    #  assignment of a layer may flip and move that layer's box / class terms by several per cent -- seen: 5.8 % on d2.loss_bbox)
    assert sorted(l0) == sorted(l1) and all(np.isfinite(v) for v in l1.values())
    assert l1['loss_occupancy'] == pytest.approx(l0['loss_occupancy'], rel=2e-2)
    assert sorted(g0) == sorted(g1) and len(g0) > 100
    assert 'transformer.decoder.layers.0.attentions.0.attn.in_proj_weight' in g1 and 'cls_branches.0.0.weight' in g1
    worst = {k: rel_l2(g1[k].cpu().numpy(), g0[k].cpu().numpy()) for k in g0}
    # Two bf16 evaluations of an untrained 3 + 6 layer network: the gradients of the sampling offsets (a sum of signed
    # trilinear slopes) move by 10-40 % between them, the typical parameter by 4 % (measured: median 0.039, max 0.42).
    # The arithmetic of the lent path itself is pinned by test_lent_linears_against_fp32 below; here only the plumbing is:
    # every gradient arrives, in fp32, on the master parameter, at the noise level of bf16.
    for k in g0:
        assert g1[k].dtype == torch.float32 and torch.isfinite(g1[k]).all(), k
        # (an almost-zero gradient may differ by more than its own norm: judged on the absolute difference then)
        assert worst[k] < 0.8 or float((g1[k] - g0[k]).abs().max()) < 1e-4 * max(1.0, float(g0[k].abs().max())), (k, worst[k])
    assert float(np.median(list(worst.values()))) < 8e-2


def test_lent_linears_against_fp32():
    """modules/lowp_params.py on a small stack of Linears (the shapes of a 64-viewpoint decoder call: 6 400 rows x 768):
    outputs, input gradient and parameter gradients of the lent bf16 path against the same stack in fp32 -- within the
    bf16 bound the package states (rel-L2 1e-2) and at least as close as autocast's own evaluation; unused parameters
    keep ``grad is None``; the multi-tensor casts are exact (round-to-nearest bf16 of the master, fp32 of the bf16 grad)."""
    lp = pkg('modules.lowp_params')
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(768, 768), torch.nn.ReLU(), torch.nn.Linear(768, 1536), torch.nn.ReLU(),
                              torch.nn.Linear(1536, 96)).to(DEV)
    spare = torch.nn.Linear(768, 32).to(DEV)                       # lent but never called
    x = torch.randn(64, 100, 768, device=DEV)
    w = torch.randn(64, 100, 96, device=DEV)

    def run(mode):
        xx = x.clone().requires_grad_(True)
        for p in list(net.parameters()) + list(spare.parameters()):
            p.grad = None
        if mode == 'fp32':
            y = net(xx)
        else:
            lent = lp.LowpParams([net, spare])
            with torch.autocast('cuda', dtype=torch.bfloat16):
                assert lent.applies(xx)
                with (lent.lent() if mode == 'lent' else contextlib.nullcontext()):
                    if mode == 'lent':
                        assert net[0].weight.dtype == torch.bfloat16 and 'forward' in net[0].__dict__
                        assert torch.equal(net[0].weight, net[0]._parameters['weight']) and net[0].weight.data_ptr() % 256 == 0
                    y = net(xx)
            assert net[0].weight.dtype == torch.float32 and 'forward' not in net[0].__dict__
        (y.float() * w).sum().backward()
        return y.detach().float(), xx.grad, [p.grad.clone() for p in net.parameters()]

    want, lent, auto = run('fp32'), run('lent'), run('autocast')
    assert spare.weight.grad is None and spare.bias.grad is None
    assert rel_l2(lent[0].cpu().numpy(), want[0].cpu().numpy()) < 1e-2
    assert rel_l2(auto[0].cpu().numpy(), want[0].cpu().numpy()) < 1e-2
    # gradients: bf16 rounding flips ReLU gates of pre-activations near zero, so BOTH bf16 evaluations sit some per cent
    # away from fp32 (6 % on the input gradient here); the lent path must not be further away than autocast's
    for g, a, wnt in zip([lent[1]] + lent[2], [auto[1]] + auto[2], [want[1]] + want[2]):
        assert g.dtype == torch.float32 and g.shape == wnt.shape
        e_lent, e_auto = rel_l2(g.cpu().numpy(), wnt.cpu().numpy()), rel_l2(a.cpu().numpy(), wnt.cpu().numpy())
        assert e_lent < 0.12 and e_lent < 1.25 * e_auto + 1e-3, (e_lent, e_auto)
    # the casts themselves
    lent_set = lp.LowpParams([net])
    with torch.autocast('cuda', dtype=torch.bfloat16), lent_set.lent():
        for m in (net[0], net[2], net[4]):
            assert torch.equal(m.weight, m._parameters['weight']) and m.weight.dtype == torch.bfloat16
        copies = [net[0].weight, net[0].bias]
        masters = [p for p in net[0].parameters()]
    # (outside the context the parameters are the fp32 masters again)
    assert all(p.dtype == torch.float32 for p in net.parameters())
    assert torch.equal(copies[0], masters[0].detach().bfloat16()) and torch.equal(copies[1], masters[1].detach().bfloat16())


def test_full_multitask_training_steps_bf16():
    """Two optimiser steps of bench.py's `--workload vocc_full_train` arithmetic (bf16 autocast, occupancy loss in row
    order, Hungarian targets batched, grad clip, fused AdamW) on 3 viewpoints: finite and decreasing-or-equal is not
    required, but every parameter that received a gradient in step 1 receives one in step 2 (no unused-parameter
    drift for DDP) and the parameters move."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    syn = pkg('synthetic')
    cfg = dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG)
    torch.manual_seed(2)
    pkg()
    head = pkg('registry').build_head(cfg)
    head.init_weights()
    for k, p in head.named_parameters():
        if k.startswith(('layout_branches.', 'query_layout_embedding.')):
            p.requires_grad_(False)
    model = bench.FullTrainer(head, 'bf16').to(DEV).train()
    B = 3
    w2p, org = syn.camera_batch(B, seed=1)
    feats = T(syn.vit_features(B, seed=100)).to(DEV).permute(1, 0, 2, 3).contiguous()
    gt = T(np.random.default_rng(7).integers(0, 17, size=(B, 504000))).to(DEV)
    gts = [syn.detection_gt(seed=40 + i, num_gt=3 + i % 5) for i in range(B)]
    gb = [T(g[0][:, :7]).to(DEV) for g in gts]
    gl = [T(g[1]).to(DEV) for g in gts]
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
    before = head.occ_proj.weight.detach().clone()
    seen = []
    for _ in range(2):
        loss = model(feats, T(w2p).to(DEV), T(org).to(DEV), gt, gb, gl)
        loss.backward()
        assert torch.isfinite(loss)
        seen.append({k for k, p in model.named_parameters() if p.grad is not None})
        torch.nn.utils.clip_grad_norm_(params, 300.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
    assert seen[0] == seen[1] and len(seen[0]) > 250
    assert float((head.occ_proj.weight.detach() - before).abs().max()) > 0


def test_layout_branch_on_gpu_matches_reference():
    """Room-layout branch (add_layout=True) end to end on the GPU: forward (head:436-532: coarse occupancy without
    upsampling + layout boxes from the decoder states) against the reference head's outputs, then the reference's
    whole ``loss_addlayout`` dict from OUR outputs."""
    from test_head_cpu import _layout_head, layout_loss_checks
    syn = pkg('synthetic')
    g = golden('layout_vocc')
    gh = golden('head_vocc')
    torch.backends.cuda.matmul.allow_tf32 = False
    head = _layout_head(DEV)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = syn.vit_features(2, seed=0)
    outs = head(T(feats[0]).to(DEV).unsqueeze(1), _metas(w2p, org, [0]))
    assert outs['all_layout_preds'].shape == (6, 1, 100, 10) and outs['occupancy_preds'].shape == (1, 7875, 16)
    assert close(outs['all_layout_preds'].detach().cpu(), g['layout_preds'], atol=2e-4, rtol=1e-4)
    assert close(outs['all_cls_scores'].detach().cpu(), gh['c3_b0_cls'], atol=2e-4, rtol=1e-4)
    assert close(outs['occupancy_preds'][0, ::5].detach().cpu(), g['occ'], atol=1e-4, rtol=1e-4)
    norm = float(outs['occupancy_preds'].detach().double().norm())
    assert abs(norm - float(g['occ_norm'])) < 1e-4 * float(g['occ_norm'])
    gt_occ, gtb, gtl, labels = layout_loss_checks(head, g, None, device=DEV)
    d = head.loss_addlayout([gtb], [T(labels).to(DEV)], [gtl], T(gt_occ).to(DEV)[None], outs)
    want = dict(zip([str(k) for k in g['dict_keys']], g['dict_vals']))
    assert sorted(d) == sorted(want)
    for k, v in want.items():
        assert float(d[k]) == pytest.approx(v, rel=1e-3, abs=1e-6), k
    sum(d.values()).backward()
    for name in ('layout_branches.5.4.weight', 'layout_branches.5.0.weight', 'occ_proj.weight'):
        grad = dict(head.named_parameters())[name].grad
        assert grad is not None and torch.isfinite(grad).all() and float(grad.abs().max()) > 0, name


def test_gpu_path_fails_loudly_without_the_hip_library(monkeypatch):
    """The dense head keeps torch formulations of its lattice algebra for CPU tensors (the fp64 algebra checks of
    tests/test_head_cpu.py run the product code on the CPU); a GPU tensor never takes them: with libver_hip.so gone the
    encoder AND the occupancy branch raise instead of computing anything with torch ops (fp32 and bf16 autocast)."""
    hip = pkg('hipops')
    syn = pkg('synthetic')
    head = _head(cases.vocc_head_cfg(), 7)
    w2p, org = syn.camera_batch(1, seed=1)
    feats = T(syn.vit_features(1, seed=0)[0]).to(DEV).unsqueeze(1)
    emb = torch.randn(1, 900, 768, device=DEV)
    monkeypatch.setattr(hip, '_lib', None)
    monkeypatch.setattr(hip, 'LIB_PATH', '/nonexistent/libver_hip.so')
    with torch.no_grad():
        with pytest.raises(hip.HipLibraryError, match='no CPU/PyTorch fallback'):
            head.lift(feats, _metas(w2p, org, [0]))
        with pytest.raises(hip.HipLibraryError):
            head.occupancy_from_volume(emb)
        with pytest.raises(hip.HipLibraryError), torch.autocast('cuda', dtype=torch.bfloat16):
            head.occupancy_from_volume(emb)
        with pytest.raises(hip.HipLibraryError):
            head.occupancy_loss(torch.randn(8192, 16, device=DEV), torch.zeros(8192, dtype=torch.long, device=DEV))


def test_graphed_head_replays_the_eager_forward_and_backward():
    """vln-ver_amd/graphs.py: the head's forward and backward recorded once as two HIP graphs and replayed -- same outputs
    and parameter gradients as the eager calls (fp32, eval mode), on inputs that differ from the ones recorded with; in
    training mode every replay draws fresh dropout seeds."""
    syn = pkg('synthetic')
    graphs = pkg('graphs')
    cfg = dict(cases.vocc_head_cfg(), train_cfg=cases.VOCC_TRAIN_CFG)
    head = _head(cfg, 7)
    for k, p in head.named_parameters():
        if k.startswith(('layout_branches.', 'query_layout_embedding.', 'positional_encoding.')):
            p.requires_grad_(False)
    w2p, org = syn.camera_batch(2, seed=1)
    feats = T(syn.vit_features(2, seed=0)).to(DEV)
    fa, fb = feats[0].unsqueeze(1).contiguous(), feats[1].unsqueeze(1).contiguous()
    wa, oa, wb, ob = (T(w2p[:1]).to(DEV), T(org[:1]).to(DEV), T(w2p[1:]).to(DEV), T(org[1:]).to(DEV))
    g = graphs.GraphedHead(head, fa, wa, oa, autocast_dtype=None, occupancy_rows=False)
    boxes, labels = cases.detection_gt(seed=41, num_gt=4)
    gt_occ = T(np.random.default_rng(3).integers(0, 17, size=(1, 504000))).to(DEV)

    def grads(outs):
        for p in head.parameters():
            p.grad = None
        sum(head.loss([T(boxes[:, :7]).to(DEV)], [T(labels).to(DEV)], gt_occ, outs).values()).backward()
        return {k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None}

    for f, w, o in ((fb, wb, ob), (fa, wa, oa)):                      # first the viewpoint the graphs were NOT recorded with
        eager = head(f, None, world2pixel=w, origin=o)
        ge = grads(eager)
        replay = g(f, w, o)
        gr = grads(replay)
        for k in ('all_cls_scores', 'all_bbox_preds', 'occupancy_preds', 'bev_embed'):
            assert close(replay[k], eager[k].float(), atol=1e-5, rtol=1e-5), k
        assert sorted(gr) == sorted(ge) and len(gr) >= 290
        for k in ge:
            d = float((gr[k] - ge[k]).abs().max())
            assert d <= 1e-5 + 1e-4 * float(ge[k].abs().max()), (k, d)
    head.train()
    # a loss of an earlier eager step that is still referenced would crash the capture (graphs.py): refused, loudly
    kept = sum(head.loss([T(boxes[:, :7]).to(DEV)], [T(labels).to(DEV)], gt_occ, eager).values())
    with pytest.raises(RuntimeError, match='still alive'):
        graphs.GraphedHead(head, fa, wa, oa)
    kept = float(kept)
    gt_ = graphs.GraphedHead(head, fa, wa, oa)                        # bf16 autocast, logits in the GEMMs' row order
    o1, o2 = gt_(fa, wa, oa), gt_(fa, wa, oa)
    assert isinstance(o1['occupancy_preds'], tuple) and o1['occupancy_preds'][0].dtype == torch.bfloat16
    c1 = o1['all_cls_scores'].clone()
    o2 = gt_(fa, wa, oa)
    assert float((o2['all_cls_scores'] - c1).abs().max()) > 0         # dropout masks differ between replays
    assert torch.isfinite(sum(head.loss([T(boxes[:, :7]).to(DEV)], [T(labels).to(DEV)], gt_occ, o2).values()))


class _LiftModel(torch.nn.Module):
    """bench.py's LiftTrainer at one micro-batch: (feats, w2p, org, gt) -> occupancy loss."""

    def __init__(self, head, autocast):
        super().__init__()
        self.head, self.autocast = head, autocast

    def forward(self, feats, w2p, org, gt):
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=self.autocast):
            emb = self.head(feats, None, only_bev=True, world2pixel=w2p, origin=org)
            return self.head.occupancy_loss_from_volume(emb, gt)


@pytest.mark.parametrize('autocast', [False, True])
def test_graphed_lift_step_replays_the_eager_training_steps(autocast):
    """graphs.GraphedLiftStep: forward + occupancy loss + backward + ``ClipAdamW`` of the lifting path as ONE hipGraph (the
    reference's samples_per_gpu = 1 operating point, vocc.py:222).  Two identical heads take the same five steps on
    alternating viewpoints -- one eagerly, one as 1 eager warm-up step + 4 replays (the inputs of replays 2 and 4 differ
    from the ones captured with): the losses of every step, the update counts, and the parameters after the five steps
    agree (the update of step k depends on AdamW's bias corrections 1 - beta^k: a replay that did not advance the
    device-side counts would scale the step by 1.9x at k = 2)."""
    import copy
    syn, graphs, opt_mod = pkg('synthetic'), pkg('graphs'), pkg('optim')
    base = _head(cases.vocc_head_cfg(), 7)
    lift = ('transformer.encoder.', 'transformer.level_embeds', 'transformer.cams_embeds', 'voxel_embedding.', 'up_sample.',
            'occ_proj.', 'occ_branches.')
    for k, p in base.named_parameters():
        p.requires_grad_(k.startswith(lift))
    heads = [base, copy.deepcopy(base)]
    before = {k: p.detach().clone() for k, p in base.named_parameters() if p.requires_grad}
    w2p, org = syn.camera_batch(2, seed=1)
    feats = T(syn.vit_features(2, seed=0)).to(DEV)
    gts = T(np.random.default_rng(3).integers(0, 17, size=(2, 1, 504000))).to(DEV)
    ins = [(feats[b].unsqueeze(1).contiguous(), T(w2p[b:b + 1]).to(DEV), T(org[b:b + 1]).to(DEV), gts[b]) for b in range(2)]
    models = [_LiftModel(h, autocast) for h in heads]                 # (eval mode: no dropout, the steps are comparable)
    opts = [opt_mod.ClipAdamW([p for p in h.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_norm=35.0)
            for h in heads]
    losses = [[], []]
    for it in range(5):                                               # eager
        opts[0].zero_grad(set_to_none=True)
        loss = models[0](*ins[it % 2])
        loss.backward()
        opts[0].step()
        losses[0].append(float(loss))
    loss = None
    step = graphs.GraphedLiftStep(models[1], opts[1], *ins[0], warmup=1)            # = step 0, eagerly, then the capture
    for it in range(1, 5):
        losses[1].append(float(step(*ins[it % 2])))
    tol = 2e-3 if autocast else 1e-5
    for a, b in zip(losses[0][1:], losses[1]):
        assert abs(a - b) <= tol * abs(a), (losses[0], losses[1])
    assert losses[0][1] != losses[0][3]                               # (the steps do move the loss: the comparison is not vacuous)
    for p in heads[1].parameters():
        if p.requires_grad:
            assert opts[1].state[p]['step'] == 5
    worst = 0.0
    for (k, p), q in zip([(k, p) for k, p in heads[0].named_parameters() if p.requires_grad],
                         [q for q in heads[1].parameters() if q.requires_grad]):
        d_e, d_g = (p.detach() - before[k]).double(), (q.detach() - before[k]).double()
        assert float(d_e.norm()) > 0
        r = float((d_e - d_g).norm() / d_e.norm())
        worst = max(worst, r)
        # AdamW's early updates are ~ lr * sign(g): an element whose tiny gradient changes sign between two runs moves by
        # 2 lr, so the bound is on the whole tensor's update, not element-wise
        assert r < (0.2 if autocast else 0.05), (k, r)
    print('graphed lifting step vs eager, %s: worst relative difference of a parameter update %.2e' % ('bf16' if autocast else 'fp32', worst))
    # an eager step after the replays: the optimizer follows the new gradient buffers and counts on
    opts[1].zero_grad(set_to_none=True)
    models[1](*ins[1]).backward()
    opts[1].step()
    assert all(opts[1].state[p]['step'] == 6 for p in heads[1].parameters() if p.requires_grad)


def test_graphed_lift_step_trains_for_many_replays():
    """Regression (round 6): on ROCm 7.2 a replayed hipGraph does not keep memset nodes in front of the kernels behind them
    unless the runtime's AQL packet capture is off (vln-ver_amd/__init__.py sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; our
    launchers zero with kernels) -- from the SECOND replay on, d(offsets) of voxels seen by two cameras and the Linears' bias
    gradients (PyTorch's multi-block reductions) came out as garbage and one non-finite element poisoned every parameter
    through the clip norm.  Train mode (dropout on), bench.py's init and optimizer, twelve replays: every gradient norm
    finite, the loss goes down, every parameter finite."""
    import os
    assert os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'
    syn, graphs, opt_mod = pkg('synthetic'), pkg('graphs'), pkg('optim')
    torch.manual_seed(2)
    head = pkg('registry').build_head(cases.vocc_head_cfg())
    head.init_weights()
    head = head.to(DEV).train()
    lift = ('transformer.encoder.', 'transformer.level_embeds', 'transformer.cams_embeds', 'voxel_embedding.', 'up_sample.',
            'occ_proj.', 'occ_branches.')
    for k, p in head.named_parameters():
        p.requires_grad_(k.startswith(lift))
    model = _LiftModel(head, True)
    params = [p for p in head.parameters() if p.requires_grad]
    opt = opt_mod.ClipAdamW(params, lr=1e-4, weight_decay=0.01, max_norm=300.0)
    w2p, org = syn.camera_batch(1, seed=1)
    feats = T(syn.vit_features(1, seed=100)).to(DEV).permute(1, 0, 2, 3).contiguous()
    gt = T(np.random.default_rng(7).integers(0, 17, size=(1, 504000))).to(DEV)
    step = graphs.GraphedLiftStep(model, opt, feats, T(w2p).to(DEV), T(org).to(DEV), gt, warmup=1)
    losses = []
    for _ in range(12):
        losses.append(float(step(*step.inputs)))
        assert bool(torch.isfinite(step.grad_norm)), losses
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert all(bool(torch.isfinite(p).all()) for p in params)
