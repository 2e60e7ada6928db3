"""``VoxelFormer``: the detector that calls the lifting path -- the class ``model = dict(type='VoxelFormer', ...)`` of
projects/configs/verformer/vocc.py names (reference: bevformer/detectors/voxelformer.py:22-400).

What the reference's detector does around ``pts_bbox_head`` is small and all of it is host code: read the six ViT
feature maps of the viewpoint from the feature store (``get_image_feature`` :317-325, CLS token dropped), take the
annotations out of ``img_metas[0]`` (:291-300), pick the head's loss entry by mode (``forward_pts_train`` :146-187) and,
at test time, decode boxes and sparse occupancy (``simple_test`` / ``simple_test_pts`` :349-391).  This mirror keeps
that surface -- constructor kwargs, ``forward(return_loss=...)``, ``forward_train(img_metas=...)``,
``forward_test`` / ``simple_test`` / ``simple_test_pts``, ``get_image_feature``, the result dicts -- and differs where
the MI355X design needs it to:

* **any number of viewpoints per call**: ``img_metas`` with B entries gives one ``[6, B, 196, 768]`` batch and one pass of
  the path (the reference reads ``img_metas[0]`` only, ``samples_per_gpu=1``); B = 1 reproduces it;
* tensors go to the module's device, not to ``.cuda()`` unconditionally (same thing on a GPU box); the features cross
  PCIe as one asynchronous copy out of a pinned staging buffer; ``autocast_dtype`` / ``occupancy_rows`` (extra
  constructor kwargs, off by default) select the bf16 training form that ``bench.py`` times;
* the image backbone / neck / point-cloud branches the config still lists are **not built**: ``forward_train`` and
  ``simple_test`` never call ``extract_feat`` in the reference either (features are precomputed, :285-289), and a DDP
  replica without the never-executed ResNet-50 + FPN needs no ``find_unused_parameters``.
  ``extract_feat`` raises with that explanation;
* ``only_det``: the reference passes three arguments to a six-argument ``loss`` there (:175-176) and cannot run; we call
  the head's ``loss_only_detection`` (head:1619), which is what the mode means.
"""
import numpy as np
import torch

from ..modules.bricks import BaseModule
from ..registry import DETECTORS, build_head
from ..volume_io import FeatureStore


def bbox3d2result(bboxes, scores, labels):
    """mmdet3d.core.bbox3d2result: the per-sample result dict, on the host."""
    to_cpu = (lambda b: b.to('cpu')) if hasattr(bboxes, 'to') else (lambda b: b)
    return dict(boxes_3d=to_cpu(bboxes), scores_3d=scores.cpu(), labels_3d=labels.cpu())


@DETECTORS.register_module(force=True)
class VoxelFormer(BaseModule):
    def __init__(self, use_grid_mask=False, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None,
                 pts_fusion_layer=None, img_backbone=None, pts_backbone=None, img_neck=None, pts_neck=None,
                 pts_bbox_head=None, img_roi_head=None, img_rpn_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, video_test_mode=False, keep_bev_history=False, use_occ_gts=True, only_occ=False,
                 only_det=False, add_layout=False, dataset_type='MP3DDataset', can_bus_in_dataset=True,
                 init_cfg=None, autocast_dtype=None, occupancy_rows=False):
        super().__init__(init_cfg)
        if pts_bbox_head is None:
            raise TypeError('VoxelFormer needs a pts_bbox_head config')
        # mmdet3d's MVXTwoStageDetector hands the ``pts`` part of train_cfg / test_cfg to the head
        head = dict(pts_bbox_head)
        head.update(train_cfg=(train_cfg or {}).get('pts') if train_cfg else None)
        head.update(test_cfg=(test_cfg or {}).get('pts') if test_cfg else None)
        self.pts_bbox_head = build_head(head)
        # listed by the config, never executed on this path (see the module docstring): kept as data
        self.unbuilt = dict(img_backbone=img_backbone, img_neck=img_neck, pts_voxel_layer=pts_voxel_layer,
                            pts_voxel_encoder=pts_voxel_encoder, pts_middle_encoder=pts_middle_encoder,
                            pts_fusion_layer=pts_fusion_layer, pts_backbone=pts_backbone, pts_neck=pts_neck,
                            img_roi_head=img_roi_head, img_rpn_head=img_rpn_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.use_grid_mask = use_grid_mask
        self.fp16_enabled = False
        self.dataset_type = dataset_type
        self.can_bus_in_dataset = can_bus_in_dataset
        self.video_test_mode = video_test_mode
        self.prev_frame_info = {'prev_bev': None, 'scene_token': None, 'prev_pos': 0, 'prev_angle': 0}
        self.keep_bev_history = keep_bev_history
        self.use_occ_gts = use_occ_gts
        self.only_occ, self.only_det, self.add_layout = only_occ, only_det, add_layout
        # how the training call runs the path on the GPU (ours, not in the reference's signature; defaults = the
        # reference's fp32 behaviour): ``autocast_dtype='bf16'`` wraps the head's forward in bf16 autocast and hands
        # fp32 outputs to the losses (what bench.py times); ``occupancy_rows`` keeps the occupancy logits in the GEMMs'
        # row order and permutes the targets instead -- same loss, no 8-GB permute
        self.autocast_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(autocast_dtype, autocast_dtype)
        self.occupancy_rows = occupancy_rows
        self._feature_stores = {}
        self._staging = {}                          # (6, B, 196, 768) -> (pinned host buffer, copy-done event)

    # ------------------------------------------------------------------ inputs
    @property
    def with_pts_bbox(self):
        return True

    def extract_feat(self, img, img_metas=None, len_queue=None):
        raise NotImplementedError(
            'VoxelFormer: the image backbone / neck are not built -- the lifting path consumes precomputed ViT '
            'features from the feature store (detectors/voxelformer.py:285-289 never calls extract_feat either)')

    def _store(self, img_ft_file):
        if img_ft_file not in self._feature_stores:
            self._feature_stores[img_ft_file] = FeatureStore(img_ft_file)
        return self._feature_stores[img_ft_file]

    def get_image_feature(self, img_ft_file, scan, viewpoint, cam_id, deg):
        """``(1, 196, 768) f32``: key ``<scan>_<vp>_i<cam_id>_<deg>`` of the feature store, CLS token dropped; every
        key is read from disk once (:317-325)."""
        return self._store(img_ft_file)._get('%s_%s_i%s_%s' % (scan, viewpoint, cam_id, deg))[:, 1:, :]

    def _device(self):
        return next(self.parameters()).device

    def viewpoint_features(self, img_metas):
        """-> ``[6, B, 196, 768]`` on the module's device: six headings at elevation 1 for every meta.  On a GPU the
        views are assembled in a pinned staging buffer (kept per batch size) and go over PCIe as ONE asynchronous copy on
        the current stream; the buffer is not touched again before that copy has finished."""
        dev = self._device()
        b = len(img_metas)
        first = self.get_image_feature(img_metas[0]['file_name'], *img_metas[0]['sample_idx'].split('_'), 1, 0)
        shape = (6, b) + tuple(first.shape[1:])
        if dev.type == 'cuda':
            host, done = self._staging.get(shape, (None, None))
            if host is None:
                host = torch.empty(shape, dtype=torch.float32, pin_memory=True)
            else:
                done.synchronize()
        else:
            host = torch.empty(shape, dtype=torch.float32)
        hv = host.numpy()
        for i, meta in enumerate(img_metas):
            scan, vp = meta['sample_idx'].split('_')
            for deg in range(6):
                hv[deg, i] = self.get_image_feature(meta['file_name'], scan, vp, 1, deg)[0]
        if dev.type != 'cuda':
            return host
        out = host.to(dev, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._staging[shape] = (host, done)
        return out

    # ------------------------------------------------------------------ training
    def forward(self, return_loss=True, **kwargs):
        return self.forward_train(**kwargs) if return_loss else self.forward_test(**kwargs)

    def forward_pts_train(self, img_feats, pts_feats, gt_bboxes_3d, gt_labels_3d, gt_layout_3d, occ_gts, flow_gts,
                          img_metas, gt_bboxes_ignore=None, prev_bev=None):
        """:146-187.  ``gt_*`` are per-sample lists; ``occ_gts[b]`` the sparse (voxel index, class) pairs."""
        head = self.pts_bbox_head
        lowp = self.autocast_dtype is not None and img_feats.is_cuda
        rows = self.occupancy_rows and not (self.only_det or self.only_occ or self.add_layout)
        # the default multi-task mode hands the head its boxes with the features: the Hungarian cost matrices then leave for the
        # host right behind the decoder and are solved there while the GPU runs the occupancy head (head.forward, targets_for)
        boxes = None
        if not (self.only_det or self.only_occ or self.add_layout):
            boxes = [head._boxes_as_tensor(b, img_feats.device) for b in gt_bboxes_3d]
        with torch.autocast('cuda', dtype=self.autocast_dtype or torch.bfloat16, enabled=lowp):
            outs = head(img_feats, img_metas, prev_bev, occupancy_rows=rows,
                        targets_for=(boxes, gt_labels_3d) if boxes is not None and img_feats.is_cuda else None)
        if lowp:
            outs = {k: (v.float() if torch.is_tensor(v) and k != 'occupancy_preds' else v) for k, v in outs.items()}
        if self.only_det:
            return head.loss_only_detection(gt_bboxes_3d, gt_labels_3d, outs)
        gt_occupancy = head.occupancy_targets(occ_gts, device=img_feats.device) if occ_gts else None
        if self.only_occ:
            return head.loss_only_occupancy(gt_bboxes_3d, gt_labels_3d, gt_occupancy, outs)
        if self.add_layout:
            return head.loss_addlayout(gt_bboxes_3d, gt_labels_3d, gt_layout_3d, gt_occupancy, outs)
        return head.loss(boxes, gt_labels_3d, gt_occupancy, outs)

    def forward_train(self, img_metas=None, **kwargs):
        """:231-315: features from the store, annotations from the metas, losses from the head."""
        img_feats = self.viewpoint_features(img_metas)
        ann = [m['ann_info'] for m in img_metas]
        gt_bboxes_3d = [a['gt_bboxes_3d'] for a in ann]
        gt_labels_3d = [a['gt_labels_3d'] for a in ann]
        gt_layout_3d = [a.get('gt_layout_3d') for a in ann]
        occ_gts = [[np.load(m['occ_gt_path'])] for m in img_metas] if self.use_occ_gts else None
        losses = dict()
        losses.update(self.forward_pts_train(img_feats, None, gt_bboxes_3d, gt_labels_3d, gt_layout_3d, occ_gts, None,
                                             img_metas, None, None))
        return losses

    # ------------------------------------------------------------------ inference
    def forward_dummy(self, img):
        return self.forward_test(img=img, img_metas=[[None]])

    def forward_test(self, img_metas, img=None, **kwargs):
        """:327-346 -> (bbox_results, occ_results); no history is kept between viewpoints."""
        self.prev_frame_info['prev_bev'] = None
        self.prev_frame_info['scene_token'] = img_metas[0]['sample_idx']
        _, bbox_results, occ_results = self.simple_test(img_metas, prev_bev=None, **kwargs)
        self.prev_frame_info['prev_pos'] = None
        self.prev_frame_info['prev_angle'] = None
        self.prev_frame_info['prev_bev'] = None
        return bbox_results, occ_results

    def simple_test(self, img_metas, img=None, prev_bev=None, rescale=False, occ_threshold=0.25):
        """:349-373 -> (volume, [dict(pts_bbox=...)] per sample, occ_results with the sparse (index, class) pairs)."""
        bbox_list = [dict() for _ in range(len(img_metas))]
        img_feats = self.viewpoint_features(img_metas)
        new_prev_bev, bbox_pts, occ_results = self.simple_test_pts(img_feats, img_metas, prev_bev, rescale=rescale)
        if occ_results['occupancy_preds'] is not None:
            occ_results = self.pts_bbox_head.get_occupancy_prediction(occ_results, occ_threshold)
        if bbox_pts is None:
            bbox_list = None
        else:
            for result_dict, pts_bbox in zip(bbox_list, bbox_pts):
                result_dict['pts_bbox'] = pts_bbox
        return new_prev_bev, bbox_list, occ_results

    def simple_test_pts(self, x, img_metas, prev_bev=None, rescale=False):
        """:376-391."""
        outs = self.pts_bbox_head(x, img_metas, prev_bev=prev_bev)
        occ_results = dict(occupancy_preds=outs.get('occupancy_preds', None), flow_preds=None)
        if outs.get('all_cls_scores') is None:                        # only_occ heads have no boxes to decode
            return outs['bev_embed'], None, occ_results
        bbox_list = self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale)
        bbox_results = [bbox3d2result(bboxes, scores, labels) for bboxes, scores, labels in bbox_list]
        return outs['bev_embed'], bbox_results, occ_results
