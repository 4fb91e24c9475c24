"""The generic quantiser family of model/quantizer.py (VectorQuantizer :33-200, EmbeddingEMA
:203-254, EMAVectorQuantizer :257-551, ProductQuantizerWrapper :554-614, get_histogram_count
:16-30) on top of the same HIP assign kernel.  They differ from model/dino_pqgo.py's Codebook in
that the *normalised* codeword is gathered (:153,:474) and in their bookkeeping outputs.
"""
import random
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .dist_utils import all_reduce_tensor

__all__ = ["VectorQuantizer", "EMAVectorQuantizer", "EmbeddingEMA", "ProductQuantizerWrapper",
           "get_histogram_count"]

_PQ_DIMS = (16, 32, 64, 128)


@torch.no_grad()
def get_histogram_count(count: torch.Tensor, prefix: str = "") -> Dict:
    """quantizer.py:16-30: the sorted-rank fractions at which cumulative usage reaches 10/50/90 %.
    One sort + cumsum + three searches on the device and a single host transfer (the reference walks
    all K entries with three tensor comparisons each, i.e. 3 K host synchronisations on a GPU)."""
    prob = count.float() / (count.sum() + 1)
    k = len(prob)
    prob, _ = torch.sort(prob, dim=0, descending=True)
    c_sum = torch.cumsum(prob, dim=0)
    thr = torch.tensor([0.1, 0.5, 0.9], device=c_sum.device, dtype=c_sum.dtype)
    hit = (c_sum.unsqueeze(0) >= thr.unsqueeze(1))
    first = torch.where(hit.any(dim=1), hit.float().argmax(dim=1), torch.full((3,), -1, device=c_sum.device))
    first = first.tolist()
    names = ("p10", "p50", "p90")
    return {f"{prefix}-{n}": (None if f < 0 else f / k) for n, f in zip(names, first)}


def _check_dim(d):
    if d not in _PQ_DIMS:
        raise ValueError(f"embed dim {d} per quantiser is not supported by the HIP assign kernel {_PQ_DIMS}")


def _norm_mode(normalize):
    if normalize in ("l2", "z_norm", "none"):
        return normalize
    if normalize == "z_trainable":
        raise NotImplementedError("normalize='z_trainable' is not supported by the HIP assign kernel")
    raise ValueError(f"Unsupported normalize type {normalize}")


def _torch_norm(z, w, normalize):
    if normalize == "l2":
        return F.normalize(z, dim=1), F.normalize(w, dim=1)
    if normalize == "z_norm":
        zs, zm = torch.std_mean(z, dim=1, keepdim=True)
        cs, cm = torch.std_mean(w, dim=1, keepdim=True)
        return (z - zm) / (zs + 1e-5), (w - cm) / (cs + 1e-5)
    return z, w


class _RestartMixin:
    @torch.no_grad()
    def prepare_restart(self, vq_current_count: torch.Tensor, z_flat: torch.Tensor) -> None:
        n_data = z_flat.shape[0]
        update_indices = torch.nonzero(vq_current_count == 0, as_tuple=True)[0]
        n_update = len(update_indices)
        z_indices = list(range(n_data))
        random.shuffle(z_indices)
        if n_update <= n_data:
            z_indices = z_indices[:n_update]
        else:
            update_indices = update_indices.tolist()
            random.shuffle(update_indices)
            update_indices = update_indices[:n_data]
        self.update_indices = update_indices
        self.update_candidates = z_flat[z_indices]


class VectorQuantizer(_RestartMixin, nn.Module):
    def __init__(self, num_codebook: int, embed_dim: int, beta: float = 0.25, normalize: Optional[str] = None,
                 use_restart: bool = False, use_gumbel: bool = False, use_split: bool = False,
                 use_weighted_sum: bool = False, update_norm: bool = True, need_initialized: str = "none") -> None:
        super().__init__()
        self.num_codebook, self.embed_dim, self.beta, self.normalize = num_codebook, embed_dim, beta, normalize
        self.codebook = nn.Embedding(num_codebook, embed_dim)
        nn.init.xavier_normal_(self.codebook.weight)
        self.register_buffer("vq_count", torch.zeros(num_codebook), persistent=False)
        self.use_restart, self.use_gumbel, self.use_split = use_restart, use_gumbel, use_split
        self.use_weighted_sum, self.update_norm, self.need_initialized = use_weighted_sum, update_norm, need_initialized
        if use_split:
            raise NotImplementedError("NOT YET implemented. Currently only for EMA.")
        if use_gumbel or use_weighted_sum:
            raise NotImplementedError("gumbel / weighted-sum quantisation has no HIP kernel")
        self.update_indices = None
        self.update_candidates = None

    @torch.no_grad()
    def restart(self) -> None:
        if (self.update_indices is not None) and (self.update_candidates is not None):
            self.codebook.weight.data[self.update_indices] = self.update_candidates
            self.vq_count.fill_(0)
            self.update_indices = None
            self.update_candidates = None

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
        """(b, d, h, w) -> (q (b,d,h,w), {loss, codebook_loss, commitment_loss, total-/current-p10/50/90},
        distance_prob (n, K))  -- quantizer.py:105-189."""
        b, d, h, w = z.shape
        _check_dim(d)
        mode = _norm_mode(self.normalize)
        rows = z.permute(0, 2, 3, 1).reshape(-1, d)
        books = self.codebook.weight.unsqueeze(0)
        n = rows.shape[0]
        output = dict()
        grad = torch.is_grad_enabled() and (rows.requires_grad or books.requires_grad)
        if not grad:
            idx, zq, loss_sum = native.pq_assign(rows.detach(), books.detach(), mode, gather_normalized=True)
            mse = loss_sum[0] / float(n * d)
            codebook_loss = commitment_loss = mse
        else:
            idx, _, _ = native.pq_assign(rows.detach(), books.detach(), mode, True, want_zq=False, want_loss=False)
            zn, cn = _torch_norm(rows, self.codebook.weight, mode)
            emb = F.embedding(idx[0], cn)
            codebook_loss = F.mse_loss(emb, zn.detach())
            commitment_loss = F.mse_loss(zn, emb.detach())
            zq = zn + (emb - zn).detach()
        prob = native.pq_prob(rows.detach(), books.detach(), mode, 1.0)
        with torch.no_grad():
            cur = torch.bincount(idx[0], minlength=self.num_codebook)
            cur = all_reduce_tensor(cur, op="sum")  # also in eval, as the reference (:162)
            self.vq_count += cur
            output.update(get_histogram_count(self.vq_count, prefix="total"))
            output.update(get_histogram_count(cur, prefix="current"))
            if self.use_restart:
                self.prepare_restart(cur, rows.detach())
        output["loss"] = codebook_loss + self.beta * commitment_loss
        output["codebook_loss"] = codebook_loss
        output["commitment_loss"] = commitment_loss
        return zq.reshape(b, h, w, d).permute(0, 3, 1, 2), output, prob


class EmbeddingEMA(nn.Module):
    def __init__(self, num_codebook: int, embed_dim: int, decay: float = 0.99, eps: float = 1e-5) -> None:
        super().__init__()
        self.decay, self.eps, self.num_codebook = decay, eps, num_codebook
        weight = torch.empty(num_codebook, embed_dim).uniform_(-1.0 / num_codebook, 1.0 / num_codebook)
        self.register_buffer("weight", weight)
        self.register_buffer("weight_avg", weight.clone())
        self.register_buffer("vq_count", torch.zeros(num_codebook))

    @torch.no_grad()
    def reset(self) -> None:
        self.weight_avg.data.copy_(self.weight.data)
        self.vq_count.fill_(0)

    def forward(self, indices: torch.Tensor) -> torch.Tensor:
        return F.embedding(indices, self.weight)

    def update(self, vq_current_count, vq_current_sum) -> None:
        """quantizer.py:233-254: EMA of counts and sums, Laplace-smoothed cluster sizes."""
        self.vq_count.data.mul_(self.decay).add_(vq_current_count, alpha=1 - self.decay)
        self.weight_avg.data.mul_(self.decay).add_(vq_current_sum, alpha=1 - self.decay)
        n = self.vq_count.sum()
        smoothed = (self.vq_count + self.eps) / (n + self.num_codebook * self.eps) * n
        self.weight.data.copy_(self.weight_avg / smoothed.unsqueeze(1))


class EMAVectorQuantizer(_RestartMixin, nn.Module):
    def __init__(self, num_codebook: int, embed_dim: int, beta: float = 0.25, normalize: Optional[str] = None,
                 decay: float = 0.99, eps: float = 1e-5, use_restart: bool = False, use_gumbel: bool = False,
                 use_split: bool = False, use_weighted_sum: bool = False, update_norm: bool = True,
                 need_initialized: str = "none") -> None:
        super().__init__()
        self.num_codebook, self.embed_dim, self.beta, self.decay = num_codebook, embed_dim, beta, decay
        self.normalize = normalize
        self.codebook = EmbeddingEMA(num_codebook, embed_dim, decay=decay, eps=eps)
        self.register_buffer("vq_count", torch.zeros(num_codebook), persistent=False)
        self.use_restart, self.use_split, self.use_gumbel = use_restart, use_split, use_gumbel
        self.update_indices = None
        self.update_candidates = None
        self.need_initialized = need_initialized
        self.use_weighted_sum, self.update_norm = use_weighted_sum, update_norm
        if use_gumbel or use_weighted_sum or use_split:
            raise NotImplementedError("gumbel / weighted-sum / split variants have no HIP kernel")

    @torch.no_grad()
    def restart(self) -> None:
        if (self.update_indices is not None) and (self.update_candidates is not None):
            self.codebook.weight.data[self.update_indices] = self.update_candidates
            self.codebook.reset()
            self.update_indices = None
            self.update_candidates = None

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
        """Flat (n, d) input -> (z_q (n,d), {loss, commitment-loss, codebook-sum[, histograms, codebook-usage]},
        distance_prob (n, K))  -- quantizer.py:383-542."""
        n, d = z.shape
        _check_dim(d)
        mode = _norm_mode(self.normalize)
        if self.need_initialized != "none" and self.need_initialized and self.training:
            if self.need_initialized == "rand":
                self.prepare_restart(torch.zeros(self.num_codebook, dtype=torch.long, device=z.device), z)
                self.restart()
            elif self.need_initialized == "kmeans":
                from sklearn.cluster import KMeans
                km = KMeans(init="k-means++", n_clusters=self.num_codebook, random_state=0)
                km.fit(z.detach().cpu().numpy())
                cent = torch.from_numpy(km.cluster_centers_).float().to(z.device)
                self.codebook.weight.data.copy_(cent)
                self.codebook.weight_avg.data.copy_(cent)
            self.need_initialized = False
        books = self.codebook.weight.unsqueeze(0)
        output = dict()
        if not (torch.is_grad_enabled() and z.requires_grad):
            idx, zq, loss_sum = native.pq_assign(z.detach(), books, mode, gather_normalized=self.update_norm)
            commitment_loss = loss_sum[0] / float(n * d)
        else:
            idx, _, _ = native.pq_assign(z.detach(), books, mode, self.update_norm, want_zq=False, want_loss=False)
            zn, cn = _torch_norm(z, self.codebook.weight, mode)
            emb = F.embedding(idx[0], cn if self.update_norm else self.codebook.weight)
            commitment_loss = F.mse_loss(zn, emb.detach())
            zq = zn + (emb - zn).detach()
        prob = native.pq_prob(z.detach(), books, mode, 1.0)
        if self.training:
            with torch.no_grad():
                one_hot = F.one_hot(idx[0], self.num_codebook).to(z.dtype)
                cur = all_reduce_tensor(one_hot.sum(dim=0), op="sum")
                cur_sum = all_reduce_tensor(one_hot.t() @ z.detach(), op="sum")
                self.vq_count += cur
                output.update(get_histogram_count(self.vq_count, prefix="total"))
                output.update(get_histogram_count(cur, prefix="current"))
                if self.use_restart:
                    self.prepare_restart(cur, z.detach())
                self.codebook.update(cur, cur_sum)
                n_dead = int((cur == 0).sum())
                output["codebook-usage"] = (self.num_codebook - n_dead) / self.num_codebook
        output["loss"] = self.beta * commitment_loss
        output["commitment-loss"] = commitment_loss
        output["codebook-sum"] = torch.sum(torch.abs(self.codebook.weight))
        return zq, output, prob


class ProductQuantizerWrapper(nn.Module):
    """quantizer.py:554-611: as the reference, usable with EMAVectorQuantizer on flat (n, D) input
    (VectorQuantizer rejects the decay/eps arguments it passes)."""

    def __init__(self, num_pq: int, num_codebook: int, embed_dim: int, beta: float = 0.25,
                 normalize: Optional[str] = None, decay: float = 0.99, eps: float = 1e-5, use_restart: bool = False,
                 use_gumbel: bool = False, use_split: bool = False, use_weighted_sum: bool = False,
                 update_norm: bool = True, need_initialized: str = "none", quantizer_cls=EMAVectorQuantizer) -> None:
        super().__init__()
        if embed_dim % num_pq != 0:
            raise ValueError(f"Embed dim {embed_dim} should be divisible by #PQ {num_pq}.")
        self.num_pq = num_pq
        self.pq_dim = embed_dim // num_pq
        self.quantizers = nn.ModuleList([
            quantizer_cls(num_codebook, self.pq_dim, beta=beta, normalize=normalize, decay=decay, eps=eps,
                          use_restart=use_restart, use_gumbel=use_gumbel, use_split=use_split,
                          use_weighted_sum=use_weighted_sum, update_norm=update_norm, need_initialized=need_initialized)
            for _ in range(self.num_pq)])

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
        z_split = torch.chunk(z, chunks=self.num_pq, dim=1)
        z_quantized, probs, outputs = [], [], dict()
        for i in range(self.num_pq):
            q_i, out_i, prob_i = self.quantizers[i](z_split[i])
            z_quantized.append(q_i)
            for k, v in out_i.items():
                outputs[k] = v if i == 0 else outputs[k] + v
            probs.append(prob_i)
        for k in outputs:
            outputs[k] = outputs[k] / self.num_pq
        return torch.cat(z_quantized, dim=1), outputs, torch.cat(probs, dim=-1)

    def extra_repr(self) -> str:
        return f"num_pq={self.num_pq}"
