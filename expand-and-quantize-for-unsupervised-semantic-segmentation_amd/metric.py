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

    def _match(self, cm_cpu: torch.Tensor):
        """One-to-one (cluster row, class column) pairs that maximise the matched pixel count (metric.py:66)."""
        rows, cols = linear_sum_assignment(cm_cpu.numpy(), maximize=True)
        return rows, cols

    def _aligned_histogram(self, cm: torch.Tensor) -> torch.Tensor:
        """The confusion matrix with its prediction rows brought into class order, square, so that the
        diagonal holds the true positives (metric.py:65-86)."""
        k = self.num_classes
        if not self.compute_hungarian:              # linear probe: predictions already are class ids
            ident = torch.arange(k).unsqueeze(1)
            self.assignments = (ident, ident.clone())
            return cm
        cm_cpu = cm.detach().cpu()
        self.assignments = self._match(cm_cpu)
        if self.extra_classes == 0:
            # row r was matched to class assignments[1][r]; sorting the rows by their class aligns them
            order = np.argsort(self.assignments[1])
            return cm[order, :]
        # more clusters than classes: every class picks its cluster (the matching of the transposed
        # problem), the clusters nobody picked are pooled into one extra row, and a zero column keeps the
        # matrix square (metric.py:70-82)
        _classes, picked = self._match(cm_cpu.t())
        matched = set(int(r) for r in self.assignments[0])
        pooled = [r for r in range(k + self.extra_classes) if r not in matched]
        rows = [cm[picked, :], cm[pooled, :].sum(dim=0, keepdim=True)]
        square = torch.cat(rows, dim=0)
        pad = torch.zeros(k + 1, 1, device=square.device)
        return torch.cat([square, pad], dim=1)

    @torch.no_grad()
    def compute(self, prefix: str = None) -> Dict[str, torch.Tensor]:
        """mIoU and pixel accuracy in percent (metric.py:60-112): the confusion matrix is summed over the
        ranks (one all-reduce), aligned, and read off its diagonal.  Classes that never occur as label or
        prediction (0 / 0) are left out of the mean, as the reference's NaN filter does (:92-93)."""
        self.confusion_matrix = all_reduce_tensor(self.confusion_matrix, op="sum")
        self.histogram = hist = self._aligned_histogram(self.confusion_matrix)
        hit = hist.diagonal()
        union = hist.sum(dim=0) + hist.sum(dim=1) - hit        # tp + fp + fn
        per_class = hit / union
        output = dict(iou=100 * per_class[~per_class.isnan()].mean(), accuracy=100 * (hit.sum() / hist.sum()))
        if self.csv_dir is not None:
            import pandas as pd
            recall = (100 * hit / hist.sum(dim=1)).unsqueeze(-1)
            os.makedirs(self.csv_dir, exist_ok=True)
            pd.DataFrame(torch.cat([hist, recall], dim=1).cpu().numpy()).to_csv(os.path.join(self.csv_dir, f"{prefix}_7.csv"))
        return output

    @torch.no_grad()
    def map_clusters(self, clusters):
        """Cluster id -> matched class id, -1 for clusters no class was matched to (metric.py:112-125).
        Needs a compute() with compute_hungarian first.  The table is laid out as the reference lays it
        out: the matched classes in cluster order, with a -1 spliced in one position AFTER each unmatched
        cluster id (or appended when that id is the current length).  With unmatched clusters only at the
        end, which is what update() produces (it drops predictions >= num_classes, :49), that is the plain
        lookup table."""
        table = [int(c) for c in self.assignments[1]]
        if self.extra_classes:
            matched = set(int(r) for r in self.assignments[0])
            for cluster in range(self.num_classes + self.extra_classes):
                if cluster not in matched:
                    table.insert(cluster if cluster == len(table) else cluster + 1, -1)
        return torch.tensor(table, dtype=torch.long)[clusters]
