"""Box (de)normalisation and NMS-free decoding ("next" row 2): same maths as the reference's
core/bbox/util.py:4-55 and core/bbox/coders/nms_free_coder.py:9-122, layout_coder.py."""
import torch

from ..registry import BBOX_CODERS


def normalize_bbox(bboxes, pc_range=None):
    """(cx,cy,cz,w,l,h,rot[,vx,vy]) -> (cx,cy,log w,log l,cz,log h,sin,cos[,vx,vy])."""
    cx, cy, cz = bboxes[..., 0:1], bboxes[..., 1:2], bboxes[..., 2:3]
    w, l, h = bboxes[..., 3:4].log(), bboxes[..., 4:5].log(), bboxes[..., 5:6].log()
    rot = bboxes[..., 6:7]
    parts = [cx, cy, w, l, cz, h, rot.sin(), rot.cos()]
    if bboxes.size(-1) > 7:
        parts += [bboxes[..., 7:8], bboxes[..., 8:9]]
    return torch.cat(parts, dim=-1)


def denormalize_bbox(nb, pc_range=None):
    rot = torch.atan2(nb[..., 6:7], nb[..., 7:8])
    cx, cy, cz = nb[..., 0:1], nb[..., 1:2], nb[..., 4:5]
    w, l, h = nb[..., 2:3].exp(), nb[..., 3:4].exp(), nb[..., 5:6].exp()
    parts = [cx, cy, cz, w, l, h, rot]
    if nb.size(-1) > 8:
        parts += [nb[:, 8:9], nb[:, 9:10]]
    return torch.cat(parts, dim=-1)


class _TopKCoder:
    def __init__(self, pc_range, voxel_size=None, post_center_range=None, max_num=100,
                 score_threshold=None, num_classes=10):
        self.pc_range = pc_range
        self.voxel_size = voxel_size
        self.post_center_range = post_center_range
        self.max_num = max_num
        self.score_threshold = score_threshold
        self.num_classes = num_classes

    def encode(self):
        pass

    def decode_single(self, cls_scores, bbox_preds):
        """cls_scores [num_query, C] logits; bbox_preds [num_query, 10] normalised."""
        cls_scores = cls_scores.sigmoid()
        scores, indexs = cls_scores.view(-1).topk(min(self.max_num, cls_scores.numel()))
        labels = indexs % self.num_classes
        bbox_index = indexs // self.num_classes
        boxes = denormalize_bbox(bbox_preds[bbox_index], self.pc_range)
        thresh = None
        if self.score_threshold is not None:
            thresh = scores > self.score_threshold
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only '
                                      'support post_center_range is not None for now!')
        rng = torch.tensor(self.post_center_range, device=scores.device)
        mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
        if thresh is not None:
            mask &= thresh
        return dict(bboxes=boxes[mask], scores=scores[mask], labels=labels[mask])

    def decode(self, preds_dicts):
        all_cls = preds_dicts['all_cls_scores'][-1]
        all_box = preds_dicts['all_bbox_preds'][-1]
        return [self.decode_single(all_cls[i], all_box[i]) for i in range(all_cls.size(0))]


@BBOX_CODERS.register_module(force=True)
class NMSFreeCoder(_TopKCoder):
    pass


@BBOX_CODERS.register_module(force=True)
class LayoutCoder(_TopKCoder):
    """core/bbox/coders/layout_coder.py: no scores -- every layout query of the last decoder layer is
    de-normalised and kept when its centre lies inside ``post_center_range``."""

    def decode_single(self, layout_preds):
        boxes = denormalize_bbox(layout_preds, self.pc_range)
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only '
                                      'support post_center_range is not None for now!')
        rng = torch.as_tensor(self.post_center_range, device=boxes.device, dtype=boxes.dtype)
        mask = (boxes[..., :3] >= rng[:3]).all(1) & (boxes[..., :3] <= rng[3:]).all(1)
        return dict(layouts=boxes[mask])

    def decode(self, preds_dicts):
        last = preds_dicts['all_layout_preds'][-1]
        return [self.decode_single(last[i]) for i in range(last.size(0))]
