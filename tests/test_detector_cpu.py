"""The ``VoxelFormer`` detector (the caller of the lifting path; reference: bevformer/detectors/voxelformer.py): host
plumbing on CPU -- the config's ``model`` dict builds, features come out of the store with the CLS token dropped,
annotations are converted the way the reference's ``loss`` does, and every mode reaches the right loss entry.  The head
itself is replaced by a recorder that returns the REFERENCE's own head outputs (tests/golden/head_vocc.npz), so the loss
values are pinned to tests/golden/loss_vocc.npz exactly as in test_head_cpu.py."""
import numpy as np
import pytest
import torch

import cases
from util import golden, pkg

T = torch.from_numpy


class BottomCentreBoxes:
    """What the dataset hands over (mmdet3d ``LiDARInstance3DBoxes``): ``tensor`` holds the BOTTOM centre,
    ``gravity_center`` lifts it by half the height."""

    def __init__(self, gravity_boxes):
        t = torch.as_tensor(gravity_boxes).clone()
        t[:, 2] = t[:, 2] - t[:, 5] * 0.5
        self.tensor = t

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = c[:, 2] + self.tensor[:, 5] * 0.5
        return c


def _store(tmp_path, feats, names):
    """feats [B, 6, 196, 768] -> a feature-store directory in the reference's key / shape convention
    (``<scan>_<vp>_i1_<deg>`` -> (1, 197, 768), CLS token first)."""
    d = tmp_path / 'vit_feats'
    d.mkdir()
    for b, name in enumerate(names):
        for deg in range(6):
            tok = np.concatenate([np.full((1, 1, 768), -7.0, np.float32), feats[b, deg][None]], 1)
            np.save(str(d / ('%s_i1_%d.npy' % (name, deg))), tok)
    return str(d)


def _metas(tmp_path, store, names, gts, occ_pairs, layouts=None):
    syn = pkg('synthetic')
    w2p, org = syn.camera_batch(len(names), seed=1)
    metas = []
    for b, name in enumerate(names):
        boxes, labels = gts[b]
        p = tmp_path / ('occ_%d.npy' % b)
        np.save(str(p), occ_pairs[b])
        ann = dict(gt_bboxes_3d=BottomCentreBoxes(boxes[:, :7]), gt_labels_3d=labels,
                   gt_layout_3d=None if layouts is None else BottomCentreBoxes(layouts[b]))
        metas.append(dict(sample_idx=name, file_name=store, ann_info=ann, occ_gt_path=str(p),
                          world2pixel=w2p[b], origin=org[b]))
    return metas


def _recorder(det, outs):
    seen = {}

    def fake(mlvl_feats, img_metas, prev_bev=None, **kw):
        seen['feats'], seen['metas'], seen['prev_bev'] = mlvl_feats, img_metas, prev_bev
        return outs
    det.pts_bbox_head.forward = fake
    return seen


def _sparse(gt_dense, classes=16):
    idx = np.nonzero(gt_dense < classes)[0]
    return np.stack([idx, gt_dense[idx]], 1).astype(np.int64)


def test_model_dict_builds_and_training_step_is_plumbed(tmp_path):
    pkg()
    cfgm = pkg('config')
    reg = pkg('registry')
    model = cfgm.load_model_cfg()                                       # the shipped entries of vocc.py: type='VoxelFormer'
    assert model['type'] == 'VoxelFormer'
    det = reg.build_detector(dict(model, img_backbone=dict(type='ResNet', depth=50), img_neck=dict(type='FPN')))
    assert type(det).__name__ == 'VoxelFormer' and det.unbuilt['img_backbone']['type'] == 'ResNet'
    assert not any(k.startswith(('img_backbone', 'img_neck')) for k in det.state_dict())    # nothing unused in a DDP replica
    assert sum(p.numel() for p in det.pts_bbox_head.parameters()) == 215991739
    assert det.pts_bbox_head.assigner is not None                       # train_cfg.pts reached the head
    with pytest.raises(NotImplementedError, match='precomputed ViT'):
        det.extract_feat(None)

    g, gh = golden('loss_vocc'), golden('head_vocc')
    syn = pkg('synthetic')
    names = ['scanA_vp0', 'scanA_vp1']
    feats = syn.vit_features(2, seed=0)
    store = _store(tmp_path, feats, names)
    gts = [cases.detection_gt(), cases.detection_gt(seed=41, num_gt=2)]
    rng = np.random.default_rng(5)
    dense = rng.integers(0, 17, size=(2, 504000))
    dense[rng.uniform(size=dense.shape) < 0.9] = 16
    metas = _metas(tmp_path, store, names, gts, [_sparse(d) for d in dense])
    logits = T((rng.standard_normal((2, 504000, 16)) * 2 - 2).astype(np.float32))

    # ---- one viewpoint: the reference's own call shape; numbers pinned to loss_vocc.npz through the reference's outputs
    outs1 = dict(all_cls_scores=T(gh['c3_b0_cls']), all_bbox_preds=T(gh['c3_b0_bbox']), occupancy_preds=logits[:1])
    seen = _recorder(det, outs1)
    losses = det(return_loss=True, img_metas=metas[:1])
    assert seen['feats'].shape == (6, 1, 196, 768) and seen['prev_bev'] is None
    assert np.array_equal(seen['feats'][:, 0].numpy(), feats[0])       # CLS token dropped, heading order kept
    assert sorted(losses) == sorted(['loss_cls', 'loss_bbox', 'loss_occupancy', 'loss_flow'] +
                                    ['d%d.loss_%s' % (i, k) for i in range(5) for k in ('cls', 'bbox')])
    assert float(losses['loss_cls']) == pytest.approx(float(g['loss_cls']), rel=1e-5)
    assert float(losses['loss_bbox']) == pytest.approx(float(g['loss_bbox']), rel=1e-5)
    head = det.pts_bbox_head
    want_occ = head.occupancy_loss(logits[:1], T(dense[:1]))
    assert float(losses['loss_occupancy']) == pytest.approx(float(want_occ), rel=1e-6)
    gt_dense = head.occupancy_targets([[np.load(metas[0]['occ_gt_path'])]])
    assert np.array_equal(gt_dense.numpy(), dense[:1])                  # sparse pairs -> dense targets, rest = "empty"

    # ---- two viewpoints in one call (our extension): one [6, 2, 196, 768] batch, per-sample annotations
    outs2 = dict(all_cls_scores=T(np.concatenate([gh['c3_b0_cls'], gh['c3_b1_cls']], 1)),
                 all_bbox_preds=T(np.concatenate([gh['c3_b0_bbox'], gh['c3_b1_bbox']], 1)), occupancy_preds=logits)
    seen = _recorder(det, outs2)
    losses2 = det.forward_train(img_metas=metas)
    assert seen['feats'].shape == (6, 2, 196, 768) and np.array_equal(seen['feats'][:, 1].numpy(), feats[1])
    want = head.loss([T(b[:, :7]) for b, _ in gts], [T(l) for _, l in gts], T(dense), outs2)
    for k in want:
        assert float(losses2[k]) == pytest.approx(float(want[k]), rel=1e-5, abs=1e-7), k

    # ---- inference: boxes decoded per sample on the host, occupancy as sparse (index, class) pairs
    seen = _recorder(det, dict(outs2, bev_embed=torch.zeros(900, 2, 768)))
    bbox_results, occ_results = det(return_loss=False, img_metas=metas)
    assert len(bbox_results) == 2 and sorted(bbox_results[0]['pts_bbox']) == ['boxes_3d', 'labels_3d', 'scores_3d']
    dec = head.get_bboxes(outs2, metas)
    for b in range(2):
        r = bbox_results[b]['pts_bbox']
        assert torch.equal(r['boxes_3d'], dec[b][0]) and torch.equal(r['scores_3d'], dec[b][1])
        assert r['boxes_3d'].shape[0] <= 50 and r['scores_3d'].device.type == 'cpu'
    sparse = head.get_occupancy_prediction(dict(occupancy_preds=logits, flow_preds=None))['occupancy_preds']
    assert torch.equal(occ_results['occupancy_preds'], sparse) and occ_results['flow_preds'] is None
    assert det.prev_frame_info['prev_bev'] is None and det.prev_frame_info['scene_token'] == 'scanA_vp0'


@pytest.mark.parametrize('mode', ['only_occ', 'only_det', 'add_layout'])
def test_detector_modes_reach_their_loss_entry(tmp_path, mode):
    pkg()
    reg = pkg('registry')
    gh = golden('head_vocc')
    syn = pkg('synthetic')
    head_cfg = cases.vocc_head_cfg(only_occ=(mode == 'only_occ'))
    if mode == 'only_det':
        head_cfg['only_det'] = True
    if mode == 'add_layout':
        head_cfg.update(add_layout=True, loss_layout=cases.LAYOUT_LOSS_CFG)
    det = reg.build_detector(dict(type='VoxelFormer', pts_bbox_head=head_cfg, train_cfg=dict(pts=cases.VOCC_TRAIN_CFG),
                                  **{mode: True}))
    head = det.pts_bbox_head
    names = ['scanB_vp3']
    feats = syn.vit_features(1, seed=2)
    store = _store(tmp_path, feats, names)
    gts = [cases.detection_gt()]
    rng = np.random.default_rng(6)
    dense = rng.integers(0, 17, size=(1, head.voxel_num))
    metas = _metas(tmp_path, store, names, gts, [_sparse(d) for d in dense], layouts=[cases.layout_gt()])
    logits = T((rng.standard_normal((1, head.voxel_num, 16)) * 2 - 2).astype(np.float32))
    cls, box = T(gh['c3_b0_cls']), T(gh['c3_b0_bbox'])
    outs = dict(all_cls_scores=None if mode == 'only_occ' else cls, all_bbox_preds=None if mode == 'only_occ' else box,
                all_layout_preds=box * 0.5 if mode == 'add_layout' else None,
                occupancy_preds=None if mode == 'only_det' else logits, bev_embed=torch.zeros(900, 1, 768))
    _recorder(det, outs)
    losses = det.forward_train(img_metas=metas)
    if mode == 'only_occ':
        assert sorted(losses) == ['loss_flow', 'loss_occupancy']
        assert float(losses['loss_occupancy']) == pytest.approx(float(head.occupancy_loss(logits, T(dense))), rel=1e-6)
        boxes_out, occ_out = det.forward_test(img_metas=metas)
        assert boxes_out is None and occ_out['occupancy_preds'].shape[1] == 2
    elif mode == 'only_det':
        assert sorted(losses) == sorted(['loss_cls', 'loss_bbox'] + ['d%d.loss_%s' % (i, k) for i in range(5) for k in ('cls', 'bbox')])
        g = golden('loss_vocc')
        assert float(losses['loss_cls']) == pytest.approx(float(g['loss_cls']), rel=1e-5)
        assert float(losses['loss_bbox']) == pytest.approx(float(g['loss_bbox']), rel=1e-5)
    else:
        want = head.loss_addlayout([T(gts[0][0][:, :7])], [T(gts[0][1])], [T(cases.layout_gt())], T(dense), outs)
        assert 'loss_layout' in losses
        for k in want:
            assert float(losses[k]) == pytest.approx(float(want[k]), rel=1e-5, abs=1e-7), k
