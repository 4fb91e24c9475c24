"""Linear and cluster probes (model/evaluator.py:11-111) with the reference's parameter names
(``linear_probe.{weight,bias}``, ``cluster_probe.clusters``).

The probes consume the hot path's output inside the eval loop (wrapper/PQGOWrapper.py:95).  The
reference up-samples the D = 1024 feature map bilinearly to the label size first (420 MB fp32 per
320x320 image).  Without autograd (the eval loop) the HIP path evaluates both probes at low
resolution and interpolates their outputs per pixel, which is the same function without that
tensor.  When gradients are needed (the probes are trained through this forward) the reference's
op sequence runs with stock device ops, in slices of a few images when no graph is recorded.
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native

__all__ = ["UnSegEvaluator", "ClusterLookup"]


class _ProbeLosses(torch.autograd.Function):
    """Both probe losses of evaluator.py:46-82 on a detached (rows, D) feature map, with the gradients of the probes'
    parameters in closed form.  The forward is one pass of the HIP probe kernels that also leaves the gradients of the
    loss sums with respect to the low-resolution class maps (``native.probe_train``); the parameter gradients are that
    table's transpose times the features -- no (B, classes, H, W) tensor and no autograd graph over pixels:
        d CE / d W = G_lin^T F / n_labelled      d CE / d b = column sums of G_lin / n_labelled
        d cluster / d c^ = G_clu^T F / (B H W)   and through c^ = c / |c|:  d c = (d c^ - c^ <c^, d c^>) / |c|"""

    @staticmethod
    def forward(ctx, rows, weight, bias, clusters, label, b, h, w):
        n_lin, n_clu = weight.shape[0], clusters.shape[0]
        probes, pbias = native.pack_probes(weight.detach(), bias.detach(), clusters.detach())
        lin, clu, losses, grad = native.probe_train(rows, probes, pbias, label, b, h, w, n_lin, n_clu)
        ctx.save_for_backward(rows, grad, losses, clusters.detach(), probes)
        ctx.meta = (n_lin, n_clu, int(label.numel()), weight.shape)
        ctx.mark_non_differentiable(lin, clu)
        return losses[0], losses[1], lin, clu

    @staticmethod
    def backward(ctx, g_lin, g_clu, _a, _b):
        rows, grad, losses, clusters, probes = ctx.saved_tensors
        n_lin, n_clu, pixels, wshape = ctx.meta
        gt_f = torch.mm(grad.t(), rows)                      # (64, D): [linear rows ; cluster rows]
        s_lin = g_lin / losses[2]
        d_w = (gt_f[:n_lin] * s_lin).reshape(wshape)
        d_b = grad[:, :n_lin].sum(0) * s_lin
        chat = probes[native.PROBE_SLOTS:native.PROBE_SLOTS + n_clu]
        d_chat = gt_f[native.PROBE_SLOTS:native.PROBE_SLOTS + n_clu] * (g_clu / float(pixels))
        norm = clusters.norm(dim=1, keepdim=True).clamp_min(1e-12)
        d_c = (d_chat - chat * (chat * d_chat).sum(dim=1, keepdim=True)) / norm
        return None, d_w, d_b, d_c, None, None, None, None


class ClusterLookup(nn.Module):
    def __init__(self, dim: int, n_classes: int):
        super().__init__()
        self.n_classes = n_classes
        self.dim = dim
        self.clusters = nn.Parameter(torch.randn(n_classes, dim))

    def forward(self, x: torch.Tensor, alpha: Optional[float] = 2.0, log_probs: bool = False):
        """evaluator.py:93-111 on a feature map already at the target resolution."""
        normed_clusters = F.normalize(self.clusters, dim=1)
        normed_features = F.normalize(x, dim=1)
        inner = torch.einsum("bchw,nc->bnhw", normed_features, normed_clusters)
        if alpha is None:
            probs = F.one_hot(torch.argmax(inner, dim=1), self.n_classes).permute(0, 3, 1, 2).contiguous().to(torch.float32)
        else:
            probs = F.softmax(inner * alpha, dim=1)
        loss = -torch.sum(probs * inner, dim=1).mean()
        if log_probs:
            return loss, F.log_softmax(inner * alpha, dim=1)
        return loss, probs


class UnSegEvaluator(nn.Module):
    def __init__(self, embed_dim: int, num_classes: int, extra_classes: int = 0, num_pq: int = 1) -> None:
        super().__init__()
        self.num_classes = num_classes
        self.linear_probe = nn.Conv2d(embed_dim, num_classes, kernel_size=1, stride=1)
        self.cluster_probe = ClusterLookup(embed_dim, num_classes + extra_classes)
        self.linear_loss = nn.CrossEntropyLoss()
        self.slice_images = 4  # images per up-sampling slice of the stock-op path without a graph
        self.use_hip = True    # False forces the stock-op path (for comparisons)
        self.hip_grads = True  # probes trained on a detached feature map: closed-form gradients from the HIP kernels

    def forward(self, out: torch.Tensor, img: torch.Tensor, label: Optional[torch.Tensor] = None,
                is_crf: bool = False):
        """evaluator.py:46-82 (is_crf=False).  Returns (linear_loss, linear_preds, cluster_loss, cluster_preds)."""
        if is_crf:
            raise NotImplementedError("dense-CRF post-processing (pydensecrf on the host) is outside this build")
        assert label is not None
        needs_graph = torch.is_grad_enabled() and (
            out.requires_grad or any(p.requires_grad for p in self.parameters()))
        if out.is_cuda and self.use_hip and not needs_graph:
            return self._forward_hip(out, label)
        if out.is_cuda and self.use_hip and not out.requires_grad:
            return self._forward_probe_grads(out, label)
        b = out.shape[0]
        step = b if (self.training and torch.is_grad_enabled()) else self.slice_images
        lin_logits, lin_preds, clu_preds, clu_inner = [], [], [], []
        for s in range(0, b, step):
            o = out[s:s + step]
            if o.shape[-2:] != label.shape[-2:]:
                o = F.interpolate(o, label.shape[-2:], mode="bilinear", align_corners=False)
            logits = self.linear_probe(o)
            lin_logits.append(logits)
            lin_preds.append(logits.argmax(1))
            # ClusterLookup with alpha=None: one-hot of the arg-max, loss = -mean(max inner product)
            normed_clusters = F.normalize(self.cluster_probe.clusters, dim=1)
            inner = torch.einsum("bchw,nc->bnhw", F.normalize(o, dim=1), normed_clusters)
            top = inner.max(dim=1)
            clu_preds.append(top.indices)
            clu_inner.append(top.values)
        linear_logits = torch.cat(lin_logits)
        linear_preds = torch.cat(lin_preds)
        cluster_preds = torch.cat(clu_preds)
        cluster_loss = -torch.cat(clu_inner).mean()
        label_flat = label.reshape(-1)
        mask = torch.logical_and(label_flat >= 0, label_flat < self.num_classes)
        logit_flat = linear_logits.permute(0, 2, 3, 1).reshape(-1, self.num_classes)
        linear_loss = self.linear_loss(logit_flat[mask], label_flat[mask]).mean()
        return linear_loss, linear_preds, cluster_loss, cluster_preds

    def _forward_hip(self, out: torch.Tensor, label: torch.Tensor):
        b, d, h, w = out.shape
        rows = out.permute(0, 2, 3, 1).reshape(b * h * w, d)  # a view when `out` is token-major underneath
        n_lin, n_clu = self.num_classes, self.cluster_probe.n_classes
        probes, bias = native.pack_probes(self.linear_probe.weight.detach(), self.linear_probe.bias.detach(),
                                          self.cluster_probe.clusters.detach())
        lin, clu, losses = native.probe_eval(rows.float(), probes, bias, label.long(), b, h, w, n_lin, n_clu)
        return losses[0], lin, losses[1], clu

    def _forward_probe_grads(self, out: torch.Tensor, label: torch.Tensor):
        """The probes are being trained (gradients flow to their parameters only; the wrapper hands
        over a detached feature map).  Same function as evaluator.py:46-82, evaluated so that autograd
        keeps (B, classes, H, W) tensors instead of the up-sampled (B, D, H, W) map: both probes are
        applied at low resolution and their class maps interpolated; the per-pixel feature norm, which
        does not depend on the parameters, comes from the HIP probe kernel."""
        b, d, h, w = out.shape
        if self.hip_grads and d % 8 == 0 and max(self.num_classes, self.cluster_probe.n_classes) <= native.PROBE_SLOTS:
            lp = self.linear_probe
            rows = out.detach().permute(0, 2, 3, 1).reshape(b * h * w, d).float()
            linear_loss, cluster_loss, lin, clu = _ProbeLosses.apply(rows, lp.weight, lp.bias, self.cluster_probe.clusters,
                                                                     label.long(), b, h, w)
            return linear_loss, lin, cluster_loss, clu
        size = label.shape[-2:]
        same = (h, w) == tuple(size)
        up = (lambda t: t) if same else (lambda t: F.interpolate(t, size, mode="bilinear", align_corners=False))
        lp = self.linear_probe
        rows = out.permute(0, 2, 3, 1).reshape(b * h * w, d).float()
        with torch.no_grad():
            probes, bias = native.pack_probes(lp.weight, lp.bias, self.cluster_probe.clusters)
            _lin, _clu, _losses, norms = native.probe_eval(rows, probes, bias, label.long(), b, h, w, self.num_classes,
                                                           self.cluster_probe.n_classes, want_norm=True)
        # (contiguous NCHW: the channels-last bilinear backward is ~5x slower, and these maps are tiny)
        maps = lambda t: t.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()   # noqa: E731
        linear_logits = up(maps(F.linear(rows, lp.weight.view(lp.out_channels, lp.in_channels), lp.bias)))
        inner = up(maps(F.linear(rows, F.normalize(self.cluster_probe.clusters, dim=1)))) / norms.unsqueeze(1)
        top = inner.max(dim=1)
        cluster_loss = -top.values.mean()
        # cross-entropy over the labelled pixels (:72-80) as log-sum-exp minus the labelled logit
        mask = torch.logical_and(label >= 0, label < self.num_classes)
        picked = linear_logits.gather(1, label.clamp(0, self.num_classes - 1).unsqueeze(1)).squeeze(1)
        ce = torch.logsumexp(linear_logits, dim=1) - picked
        linear_loss = (ce * mask).sum() / mask.sum()
        return linear_loss, linear_logits.argmax(1), cluster_loss, top.indices
