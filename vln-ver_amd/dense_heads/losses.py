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


# 0: never check the label range on the host; 1 (default): on a module's first fused call; 2: on every call.
# Whatever the setting, the kernel itself turns the loss into NaN when a label is outside [0, C] and raises a sticky
# device-side flag (ver_loss.hip) whose asynchronous host mirror (hipops.LabelRangeFlag) makes a LATER fused call
# raise -- also when the caller has cleaned the NaN away in between, as the head's nan_to_num does.
_FOCAL_CHECK = int(os.environ.get('VER_FOCAL_CHECK', '1'))


@LOSSES.register_module(force=True)
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight
        self._labels_checked = False

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        if (pred.is_cuda and weight is None and reduction == 'mean' and avg_factor is not None
                and pred.dim() == 2 and pred.size(1) % 8 == 0 and pred.size(0) >= 4096
                and pred.dtype in (torch.float32, torch.bfloat16)):
            # Fused path (ver_focal_loss_*): rows >= 4096 only, so the small detection-branch calls keep the torch
            # arithmetic; on bf16 logits the kernel uses the hardware log (tolerances: tests/test_hip_ops_gpu.py).
            # A label outside [0, C] raises in F.one_hot (and in the reference).  The kernel answers it with a NaN loss
            # and a sticky device flag that the next fused calls report without a synchronisation
            # (hipops.LabelRangeFlag; `check_labels()` below waits for it); the host-side range check with its
            # immediate, readable message costs a device->host sync and a reduction over every label, so it runs on
            # the first fused call of a module only (VER_FOCAL_CHECK=2: every call, 0: never).
            self.check_label_range(target, pred.size(1))
            from ..hipops import sigmoid_focal_loss_sum
            return self.loss_weight * (sigmoid_focal_loss_sum(pred, target, self.gamma, self.alpha) / avg_factor)
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, self.gamma, self.alpha,
                                                     reduction, avg_factor)


    def check_label_range(self, target, classes):
        """Host-side range check of the fused paths (this module's forward and the head's fused MLP + focal-loss
        Function): raises at once, like F.one_hot in the reference, on a module's first fused call
        (VER_FOCAL_CHECK=2: every call, 0: never); not under stream capture (a device -> host read)."""
        if not target.numel() or torch.cuda.is_current_stream_capturing():
            return
        if _FOCAL_CHECK >= 2 or (_FOCAL_CHECK == 1 and not self._labels_checked):
            self._labels_checked = True
            lo, hi = torch.aminmax(target)
            lo, hi = torch.stack((lo, hi)).tolist()
            if lo < 0 or hi > classes:
                raise RuntimeError('FocalLoss: target labels must be in [0, %d], got [%d, %d]' % (classes, lo, hi))

    @staticmethod
    def check_labels(device=None):
        """Wait for the device and raise if any fused focal-loss call so far saw a label outside [0, C]."""
        from ..hipops import LabelRangeFlag
        for f in list(LabelRangeFlag._per_device.values()):
            if device is None or f.dev.device == torch.device(device):
                f.poll(sync=True)


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
