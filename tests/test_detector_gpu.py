"""The ``VoxelFormer`` detector on the GPU: a training call and a test call through the reference's detector surface
(``forward(return_loss=...)`` with ``img_metas``), the lifting path underneath running on the HIP kernels.  The head
outputs seen inside the detector are pinned to the reference's (tests/golden/head_vocc.npz)."""
import numpy as np
import pytest
import torch

import cases
from test_detector_cpu import _metas, _sparse, _store
from util import close, golden, pkg

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = torch.from_numpy


def test_detector_train_and_test_calls_on_gpu(tmp_path):
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    pkg()
    syn, reg = pkg('synthetic'), pkg('registry')
    gh = golden('head_vocc')
    det = reg.build_detector(dict(type='VoxelFormer', pts_bbox_head=cases.vocc_head_cfg(),
                                  train_cfg=dict(pts=cases.VOCC_TRAIN_CFG))).eval()
    head = det.pts_bbox_head
    cw = head.code_weights.detach().clone()
    syn.load_seeded(head, 7)
    head.code_weights.data.copy_(cw)
    det.to(DEV)
    names = ['scanA_vp0', 'scanA_vp1']
    feats = syn.vit_features(2, seed=0)
    store = _store(tmp_path, feats, names)
    gts = [cases.detection_gt(seed=40 + i, num_gt=3 + i) for i in range(2)]
    dense = np.random.default_rng(9).integers(0, 17, size=(2, 504000))
    metas = _metas(tmp_path, store, names, gts, [_sparse(d) for d in dense])
    seen = {}
    hook = head.register_forward_hook(lambda m, a, out: seen.update(outs=out, feats=a[0]))
    losses = det(return_loss=True, img_metas=metas)
    outs = seen['outs']
    assert seen['feats'].is_cuda and seen['feats'].shape == (6, 2, 196, 768)
    for b in range(2):                                                     # the path under the detector == the reference's
        assert close(outs['all_cls_scores'][:, b].cpu(), gh['c3_b%d_cls' % b][:, 0], atol=2e-4, rtol=1e-4)
        assert close(outs['all_bbox_preds'][:, b].cpu(), gh['c3_b%d_bbox' % b][:, 0], atol=2e-4, rtol=1e-4)
        assert close(outs['occupancy_preds'][b, ::997].cpu(), gh['c3_b%d_occ' % b], atol=1e-4, rtol=1e-4)
    want = head.loss([T(b[:, :7]).to(DEV) for b, _ in gts], [T(l).to(DEV) for _, l in gts], T(dense).to(DEV), outs)
    assert sorted(losses) == sorted(want)
    for k in want:
        assert float(losses[k]) == pytest.approx(float(want[k]), rel=1e-4, abs=1e-7), k
    total = sum(losses.values())
    total.backward()
    g = head.transformer.encoder.layers[0].attentions[0].deformable_attention.sampling_offsets.weight.grad
    assert torch.isfinite(total) and g is not None and float(g.abs().max()) > 0
    # test call: boxes on the host, sparse occupancy pairs on the device
    with torch.no_grad():
        bbox_results, occ_results = det(return_loss=False, img_metas=metas)
    hook.remove()
    dec = head.get_bboxes(seen['outs'], metas)
    for b in range(2):
        r = bbox_results[b]['pts_bbox']
        assert r['boxes_3d'].device.type == 'cpu' and torch.equal(r['boxes_3d'], dec[b][0].cpu())
        assert r['boxes_3d'].shape[1] == 9 and r['boxes_3d'].shape[0] <= 50
    pairs = occ_results['occupancy_preds']
    assert pairs.dim() == 2 and pairs.shape[1] == 2 and int(pairs[:, 0].max()) < 2 * 504000
    assert int(pairs[:, 1].max()) < 16 and occ_results['flow_preds'] is None


def test_export_run_writes_one_volume_per_viewpoint(tmp_path):
    """projects/configs/verformer/get_occ.py: the head is built with ``getbev=<store>`` and a TEST call over the
    viewpoints leaves their volumes in the store (float64, raw (C, Z, H, W) view of the encoder output, key =
    sample_idx) -- what the VLN agent reads."""
    pkg()
    syn, reg, vio = pkg('synthetic'), pkg('registry'), pkg('volume_io')
    out_dir = str(tmp_path / 'volumes')
    det = reg.build_detector(dict(type='VoxelFormer', pts_bbox_head=dict(cases.vocc_head_cfg(), getbev=out_dir))).eval()
    syn.load_seeded(det.pts_bbox_head, 7)
    det.to(DEV)
    names = ['scanA_vp0', 'scanA_vp1']
    feats = syn.vit_features(2, seed=0)
    store = _store(tmp_path, feats, names)
    dense = np.full((2, 504000), 16)
    metas = _metas(tmp_path, store, names, [cases.detection_gt()] * 2, [_sparse(d) for d in dense])
    seen = {}
    hook = det.pts_bbox_head.register_forward_hook(lambda m, a, out: seen.update(outs=out))
    with torch.no_grad():
        det(return_loss=False, img_metas=metas)
    hook.remove()
    gh = golden('head_vocc')
    for b, name in enumerate(names):
        vol = vio.read_volume(out_dir, name)
        assert vol.dtype == np.float64 and vol.shape == (768, 4, 15, 15)
        emb = seen['outs']['bev_embed'][:, b].double().cpu().numpy()              # [Nq, C] of this viewpoint
        assert np.array_equal(vol.reshape(-1), emb.reshape(-1))                     # raw reinterpretation, head:634
        assert np.abs(emb[::7] - gh['c3_b%d_bev' % b]).max() < 1e-4                 # and it is the reference's volume
