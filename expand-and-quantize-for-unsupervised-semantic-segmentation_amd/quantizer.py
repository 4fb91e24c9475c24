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
def _histogram_ranks(counts: torch.Tensor) -> torch.Tensor:
    """counts (..., K) -> int64 (..., 3): for every count vector the first sorted rank at which the cumulative usage
    reaches 10 / 50 / 90 % (-1 where it never does).  Device tensors in, device tensor out: no host transfer."""
    prob = counts.float() / (counts.sum(dim=-1, keepdim=True) + 1)
    prob, _ = torch.sort(prob, dim=-1, descending=True)
    c_sum = torch.cumsum(prob, dim=-1)
    thr = torch.tensor([0.1, 0.5, 0.9], device=c_sum.device, dtype=c_sum.dtype)
    hit = c_sum.unsqueeze(-2) >= thr.unsqueeze(-1)                      # (..., 3, K)
    return torch.where(hit.any(dim=-1), hit.float().argmax(dim=-1), torch.full_like(hit[..., 0], -1, dtype=torch.int64))


def _histogram_dict(ranks, k: int, prefix: str) -> Dict:
    return {f"{prefix}-{n}": (None if f < 0 else f / k) for n, f in zip(("p10", "p50", "p90"), ranks)}


@torch.no_grad()
def get_histogram_count(count: torch.Tensor, prefix: str = "") -> Dict:
    """quantizer.py:16-30: the sorted-rank fractions at which cumulative usage reaches 10/50/90 %.
    One sort + cumsum + three searches on the device and a single host transfer (the reference walks
    all K entries with three tensor comparisons each, i.e. 3 K host synchronisations on a GPU)."""
    return _histogram_dict(_histogram_ranks(count).tolist(), len(count), prefix)


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
            cur = native.code_usage(idx, self.num_codebook)[0]      # (whole numbers: exact in fp32)
            cur = all_reduce_tensor(cur, op="sum")  # also in eval, as the reference (:162)
            self.vq_count += cur
            ranks = _histogram_ranks(torch.stack([self.vq_count, cur])).tolist()   # one host transfer for both
            output.update(_histogram_dict(ranks[0], self.num_codebook, "total"))
            output.update(_histogram_dict(ranks[1], self.num_codebook, "current"))
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
        output = dict()
        if self.training:
            with torch.no_grad():
                cur, cur_sum = _usage_and_sums(z.detach(), idx, self.num_codebook, d)
                ranks, n_dead = self._bookkeeping(cur[0], cur_sum[0], z.detach())
                stats = torch.cat([ranks.reshape(-1), n_dead.reshape(1)]).tolist()   # the forward's one host transfer
                output.update(_histogram_dict(stats[0:3], self.num_codebook, "total"))
                output.update(_histogram_dict(stats[3:6], self.num_codebook, "current"))
                output["codebook-usage"] = (self.num_codebook - stats[6]) / self.num_codebook
        output["loss"] = self.beta * commitment_loss
        output["commitment-loss"] = commitment_loss
        output["codebook-sum"] = torch.sum(torch.abs(self.codebook.weight))
        return zq, output, prob

    @torch.no_grad()
    def _bookkeeping(self, cur, cur_sum, z):
        """Training-mode state update from this call's (already all-reduced) usage counts (K) and per-codeword sums (K, d):
        quantizer.py:485-510.  Returns device tensors (no host transfer): histogram ranks (2, 3) and the number of
        unused codewords."""
        self.vq_count += cur
        ranks = _histogram_ranks(torch.stack([self.vq_count, cur]))
        if self.use_restart:
            self.prepare_restart(cur, z)
        self.codebook.update(cur, cur_sum)
        return ranks, (cur == 0).sum()


@torch.no_grad()
def _usage_and_sums(z, idx, num_codebook, d):
    """Usage counts (num_pq, K) and per-codeword sums of the assigned rows (num_pq, K, d) of one assign pass over
    z (n, num_pq*d) with idx (num_pq, n), from two launches (integer histogram, scatter-add) instead of a dense
    (n, K) one-hot matrix per sub-space, summed over the ranks with ONE all-reduce (the reference issues two per
    sub-quantiser, quantizer.py:490-491)."""
    cur = native.code_usage(idx, num_codebook)
    cur_sum = native.code_sums(z, idx, num_codebook, d)
    packed = all_reduce_tensor(torch.cat([cur.reshape(-1), cur_sum.reshape(-1)]), op="sum")
    return packed[:cur.numel()].view_as(cur), packed[cur.numel():].view_as(cur_sum)


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
        self.batched = True   # False: always the reference's loop over the sub-quantisers (one set of launches each)
        self.quantizers = nn.ModuleList([
            quantizer_cls(num_codebook, self.pq_dim, beta=beta, normalize=normalize, decay=decay, eps=eps,
                          use_restart=use_restart, use_gumbel=use_gumbel, use_split=use_split,
                          use_weighted_sum=use_weighted_sum, update_norm=update_norm, need_initialized=need_initialized)
            for _ in range(self.num_pq)])

    def _batched(self, z) -> bool:
        """All sub-spaces in one launch: EMA quantisers of one configuration, no gradient into z, no pending
        initialisation (the per-quantiser path handles those as the reference's loop does)."""
        q0 = self.quantizers[0]
        if not self.batched or not all(type(q) is EMAVectorQuantizer for q in self.quantizers):
            return False
        if torch.is_grad_enabled() and z.requires_grad:
            return False
        if any(q.training and q.need_initialized not in ("none", False) for q in self.quantizers):
            return False
        return z.is_cuda and all(q.normalize == q0.normalize and q.update_norm == q0.update_norm and q.beta == q0.beta and
                                 q.training == q0.training for q in self.quantizers)

    def forward(self, z: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
        if self._batched(z):
            return self._forward_batched(z)
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

    def _forward_batched(self, z):
        """quantizer.py:587-611 with the loop over sub-quantisers folded into the kernels: one assign launch and one
        probability launch for all sub-spaces (they take (n, D) rows and (num_pq, K, d) codebooks as they are), one usage
        histogram, one scatter-add and one all-reduce for the EMA statistics, one host transfer for every histogram."""
        q0 = self.quantizers[0]
        n, dim = z.shape
        d, K, num_pq = self.pq_dim, q0.num_codebook, self.num_pq
        _check_dim(d)
        mode = _norm_mode(q0.normalize)
        rows = z.detach()
        books = torch.stack([q.codebook.weight for q in self.quantizers])
        idx, zq, loss_sum = native.pq_assign(rows, books, mode, gather_normalized=q0.update_norm)
        prob = native.pq_prob(rows, books, mode, 1.0)
        commitment = loss_sum / float(n * d)                       # per sub-space
        outputs = {}
        if q0.training:
            with torch.no_grad():
                cur, cur_sum = _usage_and_sums(rows, idx, K, d)
                z_split = torch.chunk(rows, chunks=num_pq, dim=1)
                stats = []
                for i, q in enumerate(self.quantizers):
                    ranks, n_dead = q._bookkeeping(cur[i], cur_sum[i], z_split[i])
                    stats.append(torch.cat([ranks.reshape(-1), n_dead.reshape(1)]))
                stats = torch.stack(stats).tolist()                # the forward's one host transfer
            for i in range(num_pq):                                # the mean over sub-quantisers, summed in the reference's order
                out_i = dict(_histogram_dict(stats[i][0:3], K, "total"))
                out_i.update(_histogram_dict(stats[i][3:6], K, "current"))
                out_i["codebook-usage"] = (K - stats[i][6]) / K
                for k, v in out_i.items():
                    outputs[k] = v if i == 0 else outputs[k] + v
            books = torch.stack([q.codebook.weight for q in self.quantizers])   # (the EMA update has changed them)
        sums = books.abs().sum(dim=(1, 2))
        for i in range(num_pq):
            out_i = {"loss": q0.beta * commitment[i], "commitment-loss": commitment[i], "codebook-sum": sums[i]}
            for k, v in out_i.items():
                outputs[k] = v if (i == 0 or k not in outputs) else outputs[k] + v
        for k in outputs:
            outputs[k] = outputs[k] / num_pq
        # prob comes out as (n, num_pq * K): the concatenation over sub-spaces the reference builds with torch.cat (:611)
        return zq, outputs, prob

    def extra_repr(self) -> str:
        return f"num_pq={self.num_pq}"
