"""Confusion-matrix metrics of the eval loop (model/metric.py:13-124): ``UnSegMetrics`` with the
reference's constructor, ``reset / update / compute / map_clusters`` and result keys.

Semantics kept from the reference:
  * a pixel counts only when both label and prediction lie in [0, num_classes)  (:49)
  * confusion[pred, label] accumulates as int64 on the device                     (:53-58)
  * compute() sums the matrix over ranks, then for the cluster probe finds the one-to-one
    cluster -> class assignment that maximises the matched pixels (Hungarian, scipy) (:63-82);
    the linear probe keeps classes in order (:83-86)
  * mIoU skips classes whose tp + fp + fn is zero (NaN) and is reported in percent (:92-97)
The reference also writes a per-class CSV into a hard-coded ``./class_matrix/...`` directory on every
compute() (:100-110); here that is opt-in through ``csv_dir``.
"""
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment

from . import native
from .dist_utils import all_reduce_tensor

__all__ = ["UnSegMetrics"]


class UnSegMetrics(nn.Module):
    def __init__(self, num_classes: int, extra_classes: int, compute_hungarian: bool, device: torch.device,
                 csv_dir: Optional[str] = None) -> None:
        super().__init__()
        if (not compute_hungarian) and extra_classes != 0:
            raise ValueError("No hungarian means that all classes are in order, so extra classes should be 0.")
        self.num_classes = num_classes
        self.extra_classes = extra_classes
        self.compute_hungarian = compute_hungarian
        self.device = device
        self.csv_dir = csv_dir
        self.register_buffer("confusion_matrix",
                             torch.zeros(num_classes + extra_classes, num_classes, dtype=torch.long, device=device))
        self.assignments = None
        self.histogram = None

    def reset(self):
        self.confusion_matrix.fill_(0)
        self.assignments = None
        self.histogram = None

    @torch.no_grad()
    def update(self, preds: torch.Tensor, label: torch.Tensor):
        """Accumulate the confusion matrix (metric.py:45-58), no host sync: one HIP kernel on the GPU,
        one bincount over label * n_pred + pred on CPU tensors."""
        if self.confusion_matrix.is_cuda and preds.is_cuda and label.is_cuda:
            native.confusion_update(self.confusion_matrix, preds, label, self.num_classes)
            return
        preds = preds.reshape(-1)
        label = label.reshape(-1)
        n_pred = self.num_classes + self.extra_classes
        keep = (label >= 0) & (label < self.num_classes) & (preds >= 0) & (preds < self.num_classes)
        # rejected pixels go to one overflow bin instead of a data-dependent boolean gather
        key = torch.where(keep, label * n_pred + preds, torch.full_like(label, self.num_classes * n_pred))
        counts = torch.bincount(key, minlength=self.num_classes * n_pred + 1)[:-1]
        self.confusion_matrix += counts.reshape(self.num_classes, n_pred).t()

    @torch.no_grad()
    def compute(self, prefix: str = None) -> Dict[str, torch.Tensor]:
        """mIoU and pixel accuracy in percent (metric.py:60-112)."""
        self.confusion_matrix = all_reduce_tensor(self.confusion_matrix, op="sum")
        cm = self.confusion_matrix
        if self.compute_hungarian:
            self.assignments = linear_sum_assignment(cm.detach().cpu(), maximize=True)
            if self.extra_classes == 0:
                self.histogram = cm[np.argsort(self.assignments[1]), :]
            else:
                assignments_t = linear_sum_assignment(cm.detach().cpu().t(), maximize=True)
                histogram = cm[assignments_t[1], :]
                missing = list(set(range(self.num_classes + self.extra_classes)) - set(self.assignments[0]))
                histogram = torch.cat([histogram, cm[missing, :].sum(0, keepdim=True)], dim=0)
                new_col = torch.zeros(self.num_classes + 1, 1, device=histogram.device)
                self.histogram = torch.cat([histogram, new_col], dim=1)
        else:
            ar = torch.arange(self.num_classes).unsqueeze(1)
            self.assignments = (ar, ar.clone())
            self.histogram = cm

        tp = torch.diag(self.histogram)
        fp = torch.sum(self.histogram, dim=0) - tp
        fn = torch.sum(self.histogram, dim=1) - tp
        iou = tp / (tp + fp + fn)
        iou = iou[~torch.isnan(iou)].mean()
        accuracy = torch.sum(tp) / torch.sum(self.histogram)
        output = dict(iou=100 * iou, accuracy=100 * accuracy)

        if self.csv_dir is not None:
            import pandas as pd
            precision = (100 * tp / (tp + fn)).unsqueeze(-1)
            os.makedirs(self.csv_dir, exist_ok=True)
            pd.DataFrame(torch.cat([self.histogram, precision], dim=1).cpu().numpy()).to_csv(
                os.path.join(self.csv_dir, f"{prefix}_7.csv"))
        return output

    @torch.no_grad()
    def map_clusters(self, clusters):
        """Cluster id -> matched class id, -1 for unmatched extra clusters (metric.py:114-124)."""
        if self.extra_classes == 0:
            return torch.tensor(self.assignments[1])[clusters]
        missing = sorted(set(range(self.num_classes + self.extra_classes)) - set(self.assignments[0]))
        cluster_to_class = self.assignments[1]
        for entry in missing:
            if entry == cluster_to_class.shape[0]:
                cluster_to_class = np.append(cluster_to_class, -1)
            else:
                cluster_to_class = np.insert(cluster_to_class, entry + 1, -1)
        return torch.tensor(cluster_to_class)[clusters]
