"""Losses named by vocc.py:182-195 ("next" row 2 of SURVEY.md 8f), restated from the published
behaviour of mmdet 2.14.0 (mmdet is not vendored by the reference).
The [N,16] occupancy focal loss (N = 504 000 x viewpoints) runs on the fused HIP kernels
(``ver_focal_loss_*``) when it is on the GPU; small / weighted cases are elementwise torch ops."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import LOSSES


def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean',
                       avg_factor=None):
    """pred [N,C] logits; target int64 [N] in [0,C] where C = background (all-zero one-hot)."""
    num_classes = pred.size(1)
    t = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    focal = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * focal
    if weight is not None and weight.dim() == 1 and loss.dim() == 2:
        weight = weight.view(-1, 1)
    return _reduce(loss, weight, reduction, avg_factor)


_FOCAL_CHECK = os.environ.get('VER_FOCAL_CHECK', '1') != '0'


@LOSSES.register_module(force=True)
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        if (pred.is_cuda and weight is None and reduction == 'mean' and avg_factor is not None
                and pred.dim() == 2 and pred.size(1) % 8 == 0 and pred.size(0) >= 4096
                and pred.dtype in (torch.float32, torch.bfloat16)):
            # Fused path (ver_focal_loss_*): rows >= 4096 only, so the small detection-branch calls keep the torch
            # arithmetic; on bf16 logits the kernel uses the hardware log (tolerances: tests/test_hip_ops_gpu.py).
            # The kernel compares `target == class`, so a label outside [0, C] would silently count as background
            # where F.one_hot (and the reference) raise: check once per call.  VER_FOCAL_CHECK=0 skips the check (it
            # costs one device->host sync).
            if _FOCAL_CHECK and target.numel():
                lo, hi = int(target.min()), int(target.max())
                if lo < 0 or hi > pred.size(1):
                    raise RuntimeError('FocalLoss: target labels must be in [0, %d], got [%d, %d]' % (pred.size(1), lo, hi))
            from ..hipops import sigmoid_focal_loss_sum
            return self.loss_weight * (sigmoid_focal_loss_sum(pred, target, self.gamma, self.alpha) / avg_factor)
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, self.gamma, self.alpha,
                                                     reduction, avg_factor)


@LOSSES.register_module(force=True)
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        if target.numel() == 0:
            return pred.sum() * 0
        return self.loss_weight * _reduce((pred - target).abs(), weight, reduction, avg_factor)


@LOSSES.register_module(force=True)
class GIoULoss(nn.Module):
    """Configured with loss_weight=0.0 in vocc.py:189 ("fake" entry kept for the DETR head
    contract); never evaluated on the VER path."""

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, **kwargs):
        raise NotImplementedError('GIoULoss is a zero-weight placeholder in vocc.py')
