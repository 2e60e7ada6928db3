"""Occupancy post-processing metrics ("next" row 4 of SURVEY.md 8f): confusion-matrix IoU / mIoU
as the reference's datasets/occupancy_metrics.py:3-90 (``SSCMetrics``; class ``n_classes-1``... the
LAST row/column of the histogram is the empty class)."""
import numpy as np


class SSCMetrics:
    def __init__(self, n_classes=17):
        self.n_classes = n_classes
        self.empty_label = n_classes
        self.hist = np.zeros((n_classes, n_classes))

    @staticmethod
    def hist_info(n_cl, pred, gt):
        """rows = reference label, cols = prediction; labels outside [0, n_cl) are ignored."""
        assert pred.shape == gt.shape
        k = (gt >= 0) & (gt < n_cl)
        hist = np.bincount(n_cl * gt[k].astype(int) + pred[k].astype(int), minlength=n_cl ** 2)
        return hist.reshape(n_cl, n_cl), int(np.sum(pred[k] == gt[k])), int(np.sum(k))

    def add_batch(self, y_pred, y_true, visible_mask=None):
        y_pred, y_true = np.asarray(y_pred).flatten(), np.asarray(y_true).flatten()
        if visible_mask is not None:
            keep = np.asarray(visible_mask).flatten() == 1
            y_pred, y_true = y_pred[keep], y_true[keep]
        self.hist = self.hist + self.hist_info(self.n_classes, y_pred, y_true)[0]

    def get_stats(self):
        d = np.diag(self.hist)
        miou = d / (self.hist.sum(1) + self.hist.sum(0) - d + 1e-6) * 100.0
        tp = np.sum(self.hist[:-1, :-1])
        fp = np.sum(self.hist[-1, :-1])
        fn = np.sum(self.hist[:-1, -1])
        if tp != 0:
            precision, recall, iou = tp / (tp + fp), tp / (tp + fn), tp / (tp + fp + fn) * 100.0
        else:
            precision, recall, iou = 0, 0, 0
        iou_ssc = miou[:self.n_classes - 1]
        return dict(iou=iou, precision=precision, recall=recall, iou_ssc=iou_ssc, miou=np.mean(iou_ssc))

    def reset(self):
        self.hist = np.zeros((self.n_classes, self.n_classes))


def dense_labels(sparse_pred, num_voxels, empty_label):
    """(index, class) pairs of ``get_occupancy_prediction`` -> dense label vector."""
    out = np.full(num_voxels, empty_label, dtype=np.int64)
    sp = np.asarray(sparse_pred)
    out[sp[:, 0]] = sp[:, 1]
    return out
