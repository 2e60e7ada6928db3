"""Hungarian target assignment for the detection branch ("next" row 2 of SURVEY.md 8f).

``HungarianAssigner3D`` / ``BBox3DL1Cost`` follow the reference's
core/bbox/assigners/hungarian_assigner_3d.py:16-143 and core/bbox/match_costs/match_cost.py:5-27;
``FocalLossCost`` / ``IoUCost`` / the pseudo sampler are restated from mmdet 2.14.0 (SURVEY.md
B.12).  The matching itself stays on the host (scipy ``linear_sum_assignment`` on a 100 x G cost
matrix), as in the reference."""
import torch

from ..registry import BBOX_ASSIGNERS, MATCH_COST, build_from_cfg
from .coders import normalize_bbox

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:                                   # pragma: no cover
    linear_sum_assignment = None


@MATCH_COST.register_module(force=True)
class FocalLossCost:
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module(force=True)
class BBox3DL1Cost:
    def __init__(self, weight=1.):
        self.weight = weight

    def __call__(self, bbox_pred, gt_bboxes):
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MATCH_COST.register_module(force=True)
class IoUCost:
    """"Fake cost" of vocc.py:204 (weight 0.0): built for the DETR head contract, never called."""

    def __init__(self, iou_mode='giou', weight=1.):
        self.weight, self.iou_mode = weight, iou_mode


class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult:
    """mmdet PseudoSampler.sample: every assigned query is a positive, every 0 a negative."""

    def __init__(self, assign_result, bboxes, gt_bboxes):
        self.pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        self.neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        self.pos_assigned_gt_inds = assign_result.gt_inds[self.pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = gt_bboxes.new_zeros((0, gt_bboxes.shape[-1] if gt_bboxes.dim() > 1 else 4))
        else:
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]


@BBOX_ASSIGNERS.register_module(force=True)
class HungarianAssigner3D:
    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.),
                 reg_cost=dict(type='BBoxL1Cost', weight=1.0), iou_cost=dict(type='IoUCost', weight=0.0),
                 pc_range=None):
        self.cls_cost = build_from_cfg(cls_cost, MATCH_COST)
        self.reg_cost = build_from_cfg(reg_cost, MATCH_COST)
        self.iou_cost = build_from_cfg(iou_cost, MATCH_COST)
        self.pc_range = pc_range

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_bboxes_ignore=None, layout=False, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        reg_cost = self.reg_cost(bbox_pred[:, :8], normalize_bbox(gt_bboxes, self.pc_range)[:, :8])
        cost = reg_cost if layout else self.cls_cost(cls_pred, gt_labels) + reg_cost
        if linear_sum_assignment is None:
            raise ImportError('Please run "pip install scipy" to install scipy first.')
        rows, cols = linear_sum_assignment(cost.detach().float().cpu())
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels if gt_labels.dim() < 1 else gt_labels[cols]
        return AssignResult(num_gts, gt_inds, None, labels=labels)


def build_assigner(cfg):
    return build_from_cfg(cfg, BBOX_ASSIGNERS)
