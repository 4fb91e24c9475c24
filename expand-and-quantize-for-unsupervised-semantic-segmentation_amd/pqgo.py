"""EQUSS model: frozen DINO features -> expansion head -> product quantisation, with the module
tree, constructor config, forward signature and return tuple of the reference's
model/dino_pqgo.py (DIONPQGO :15-154, Codebook :460-705, ProductQuantizerWrapper :708-776).

Inference (the timed path) runs entirely in the HIP kernels: ViT -> head GEMMs -> PQ assign, all
token-major, no layout copies.  Training keeps the frozen ViT and the nearest-codeword search in
HIP and uses autograd-visible elementwise torch ops only for what needs gradients (head, losses).
"""
import random
import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .dist_utils import all_reduce_tensor
from .featurizer import DinoFeaturizer

__all__ = ["DIONPQGO", "Codebook", "ProductQuantizerWrapper", "STEGOLoss"]

_PQ_DIMS = (16, 32, 64, 128)


# ------------------------------------------------------------------------------------------------
# STEGO correspondence loss (model/loss.py:647-739); training only, tiny (11x11 samples), torch ops.
# ------------------------------------------------------------------------------------------------
def _corr(a, b):
    return torch.einsum("nchw,ncij->nhwij", a, b)


def _unit(t):
    return F.normalize(t, dim=1, eps=1e-10)


class _BilinearGather(torch.autograd.Function):
    """out[p] = sum_k wgt[p, k] * rows[idx[p, k]]: the four-neighbour gather behind the point sampling
    below.  The backward scatters into ONE dense gradient (a zero fill and four index_add_), where
    autograd through grid_sample or through advanced indexing would build a dense gradient per call."""

    @staticmethod
    def forward(ctx, rows, idx, wgt):
        ctx.save_for_backward(idx, wgt)
        ctx.rows_shape = rows.shape
        if rows.is_cuda and rows.dtype == torch.float32 and rows.shape[1] % 4 == 0:
            return native.tap_gather(rows, idx, wgt)            # one kernel for the four taps
        out = rows[idx[:, 0]] * wgt[:, 0:1]
        for k in range(1, 4):
            out.addcmul_(rows[idx[:, k]], wgt[:, k:k + 1])
        return out

    @staticmethod
    def backward(ctx, g):
        idx, wgt = ctx.saved_tensors
        if g.is_cuda and g.dtype == torch.float32:
            return native.tap_scatter(g, idx, wgt, ctx.rows_shape[0]), None, None
        grad = g.new_zeros(ctx.rows_shape)
        for k in range(4):
            grad.index_add_(0, idx[:, k], g * wgt[:, k:k + 1])
        return grad, None, None


def _sample_taps(coords, h, w, batch_index=None):
    """The four (token row, weight) taps of every sample point: coords (S, B, n, n, 2) in [-1, 1] on an (h, w) map,
    batch_index (S, B) long or None (set s samples image batch_index[s, b], the reference's ``t[perm]``).  Returns
    idx (S*B*n*n, 4) long, wgt (S*B*n*n, 4); shared by every tensor of that spatial size."""
    s_, b = coords.shape[:2]
    g = coords.permute(0, 1, 3, 2, 4)                      # sample(): grid = coords.permute(0, 2, 1, 3)
    ix = (((g[..., 0] + 1) / 2) * (w - 1)).clamp(0, w - 1)
    iy = (((g[..., 1] + 1) / 2) * (h - 1)).clamp(0, h - 1)
    x0, y0 = ix.floor(), iy.floor()
    x1, y1 = x0 + 1, y0 + 1
    wgt = torch.stack(((x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)), dim=-1)
    x0l, y0l = x0.long(), y0.long()
    x1l, y1l = (x0l + 1).clamp_(max=w - 1), (y0l + 1).clamp_(max=h - 1)   # weight is 0 where the neighbour is outside
    if batch_index is None:
        batch_index = torch.arange(b, device=coords.device).expand(s_, b)
    base = (batch_index * (h * w)).reshape(s_, b, 1, 1)
    idx = torch.stack((base + y0l * w + x0l, base + y0l * w + x1l, base + y1l * w + x0l, base + y1l * w + x1l), dim=-1)
    return idx.reshape(-1, 4), wgt.reshape(-1, 4)


def _sample_points(t, coords, batch_index=None, taps=None):
    """The reference's ``sample(t, coords)`` (model/loss.py:647-648: grid_sample, bilinear, border
    padding, align_corners=True) for S coordinate sets at once.

    t (B, C, h, w); coords (S, B, n, n, 2) in [-1, 1]; batch_index (S, B) long or None: set s samples
    image batch_index[s, b] (the reference's ``t[perm]``) without copying t.  Returns (S, B, C, n, n).
    A sample point touches four tokens, so this is a row gather from the token-major tensor (a free
    view for the tensors of the training pass) instead of a kernel that walks, and in the backward
    zero-fills and atomically updates, the whole (B, C, h, w) map once per coordinate set.
    ``taps`` = the result of _sample_taps for these coordinates, when the caller has it already."""
    s_, b, n1, n2, _ = coords.shape
    _, c, h, w = t.shape
    rows = _rows_of(t)
    idx, wgt = taps if taps is not None else _sample_taps(coords, h, w, batch_index)
    out = _BilinearGather.apply(rows, idx, wgt.to(rows.dtype))
    return out.reshape(s_, b, n2, n1, c).permute(0, 1, 4, 2, 3)


def _derangement_like(size: int, device):
    """loss.py:659-663: a random permutation with fixed points bumped by one."""
    perm = torch.randperm(size, device=device, dtype=torch.long)
    # (the reference's masked `perm[perm == arange] += 1`, written without boolean indexing: that needs the number of
    # fixed points on the host, i.e. two device synchronisations per permutation)
    return torch.where(perm == torch.arange(size, device=device), perm + 1, perm) % size


class STEGOLoss(nn.Module):
    def __init__(self, cfg: dict):
        super().__init__()
        self.cfg = cfg
        # (coords1, coords2, perms) to use in place of the random draws of the next forward calls
        # (loss.py:709-710,:725): lets a test replay the draws recorded from another implementation
        self.replay = None

    def helper(self, f1, f2, c1, c2, shift):
        with torch.no_grad():
            fd = _corr(_unit(f1), _unit(f2))
            if self.cfg["pointwise"]:
                old_mean = fd.mean()
                fd -= fd.mean([3, 4], keepdim=True)
                fd = fd - fd.mean() + old_mean
        cd = _corr(_unit(c1), _unit(c2))
        lo = 0.0 if self.cfg["zero_clamp"] else -9999.0
        if self.cfg["stabilize"]:
            loss = -cd.clamp(lo, .8) * (fd - shift)
        else:
            loss = -cd.clamp(lo) * (fd - shift)
        return loss, cd

    def forward(self, orig_feats, orig_feats_pos, orig_code, orig_code_pos):
        n = self.cfg["feature_samples"]
        b = orig_feats.shape[0]
        dev = orig_feats.device
        shape = [b, n, n, 2]
        if self.replay is not None:
            coords1, coords2, perms = self.replay
            coords1, coords2 = coords1.to(dev, torch.float32), coords2.to(dev, torch.float32)
            perms = [p.to(dev, torch.long) for p in perms]
            if list(coords1.shape) != shape or list(coords2.shape) != shape or len(perms) != self.cfg["neg_samples"]:
                raise ValueError("STEGOLoss.replay does not match the batch / feature_samples / neg_samples in use")
        else:
            coords1 = torch.rand(shape, device=dev) * 2 - 1
            coords2 = torch.rand(shape, device=dev) * 2 - 1
            # same random stream as the reference (loss.py:694-731: coords1, coords2, then one permutation
            # per negative); all point sets of one tensor are then gathered in a single call
            perms = [_derangement_like(b, dev) for _ in range(self.cfg["neg_samples"])]
        own = torch.arange(b, device=dev)
        coords = torch.stack([coords1] + [coords2] * len(perms))
        which = torch.stack([own] + perms)
        # feature and code maps have the same spatial size: their sample points share the taps
        same = orig_feats.shape[-2:] == orig_code.shape[-2:]
        taps_all = _sample_taps(coords, orig_code.shape[-2], orig_code.shape[-1], which) if same else None
        taps_pos = _sample_taps(coords2[None], orig_code.shape[-2], orig_code.shape[-1]) if same else None
        with torch.no_grad():
            feats_all = _sample_points(orig_feats, coords, which, taps_all)
            feats_pos = _sample_points(orig_feats_pos, coords2[None], None, taps_pos)[0]
        code_all = _sample_points(orig_code, coords, which, taps_all)
        code_pos = _sample_points(orig_code_pos, coords2[None], None, taps_pos)[0]
        feats, code = feats_all[0], code_all[0]
        # All helper() calls of loss.py:733-739 share their first operands: one batched correlation per tensor over the
        # stacked second operands [intra, inter, negatives...] instead of 2 + neg_samples passes of ~25 small launches.
        shifts = [self.cfg["pos_intra_shift"], self.cfg["pos_inter_shift"]] + [self.cfg["neg_inter_shift"]] * len(perms)
        loss = self.helper_sets(feats, torch.cat((feats_all[:1], feats_pos[None], feats_all[1:])),
                                code, torch.cat((code_all[:1], code_pos[None], code_all[1:])),
                                torch.tensor(shifts, device=dev, dtype=feats.dtype))
        return (self.cfg["pos_intra_weight"] * loss[0].mean() + self.cfg["pos_inter_weight"] * loss[1].mean() +
                self.cfg["neg_inter_weight"] * loss[2:].mean())

    def helper_sets(self, f1, f2_sets, c1, c2_sets, shifts):
        """helper() for S second operands at once: f2_sets (S, B, C, n, n), c2_sets (S, B, D, n, n), shifts (S,) ->
        loss (S, B, n, n, n, n).  Same arithmetic per set as helper(); the means run over each set's own tensor."""
        with torch.no_grad():
            fd = torch.einsum("nchw,sncij->snhwij", _unit(f1), F.normalize(f2_sets, dim=2, eps=1e-10))
            if self.cfg["pointwise"]:
                old_mean = fd.mean(dim=(1, 2, 3, 4, 5), keepdim=True)
                fd -= fd.mean([4, 5], keepdim=True)
                fd = fd - fd.mean(dim=(1, 2, 3, 4, 5), keepdim=True) + old_mean
        cd = torch.einsum("nchw,sncij->snhwij", _unit(c1), F.normalize(c2_sets, dim=2, eps=1e-10))
        lo = 0.0 if self.cfg["zero_clamp"] else -9999.0
        cd = cd.clamp(lo, .8) if self.cfg["stabilize"] else cd.clamp(lo)
        return -cd * (fd - shifts.view(-1, 1, 1, 1, 1, 1))


# ------------------------------------------------------------------------------------------------
# Expansion head in the training pass
# ------------------------------------------------------------------------------------------------
class _HeadFunction(torch.autograd.Function):
    """cluster1(f) + cluster2(f) on token-major rows: forward through the same HIP head GEMMs as the
    eval path (bf16 operands, fp32 accumulation and output), backward as bf16 GEMMs with fp32
    accumulation on the HIP kernels (``native.gemm_tn_bf16`` for the weight gradients, ``native.gemm_bf16`` for the
    gradient through the second branch) -- the precision class of the reference's training pass, which runs these
    convolutions under fp16 autocast (train.py:80).  The frozen backbone needs no input gradient.
    The HIP forward leaves [f | relu(W2a f + b2a)] behind in bf16, which is what the backward needs."""

    @staticmethod
    def forward(ctx, rows, w1, b1, w2a, b2a, w2b, b2b):
        n, c = rows.shape
        d = w1.shape[0]
        fh = torch.empty((n, 2 * c), dtype=torch.bfloat16, device=rows.device)
        fh[:, :c] = rows
        w_cat = torch.cat((w1.reshape(d, c), w2b.reshape(d, c)), dim=1).to(torch.bfloat16).contiguous()
        code = native.expand_head(fh, w2a.reshape(c, c).to(torch.bfloat16).contiguous(), b2a.float().contiguous(),
                                  w_cat, (b1 + b2b).float().contiguous(), d)
        ctx.save_for_backward(fh, w_cat)
        ctx.shapes = (w1.shape, w2a.shape, w2b.shape)
        return code

    native_backward = True   # False: the same products as torch.mm calls (for comparisons)

    @staticmethod
    def backward(ctx, g):
        fh, w_cat = ctx.saved_tensors
        c = fh.shape[1] // 2
        d = w_cat.shape[0]
        s1, s2a, s2b = ctx.shapes
        g_bf = g.to(torch.bfloat16)
        if _HeadFunction.native_backward and c % 128 == 0 and d % 128 == 0:
            # dW = g^T f sums over batch * tokens: the split-K transposed-operand kernel; d_pre is an ordinary A W^T GEMM
            dw_cat = native.gemm_tn_bf16(g_bf, fh)                   # (D, 2C) = [dW1 | dW2b]
            d_pre = native.gemm_bf16(g_bf, w_cat[:, c:].t().contiguous(), None, native.EPI_BF16) * (fh[:, c:] > 0)
            dw2a = native.gemm_tn_bf16(d_pre, fh[:, :c])
        else:
            dw_cat = torch.mm(g_bf.t(), fh).float()                  # (D, 2C) = [dW1 | dW2b]
            d_pre = torch.mm(g_bf, w_cat[:, c:]) * (fh[:, c:] > 0)   # through the ReLU
            dw2a = torch.mm(d_pre.t(), fh[:, :c]).float()
        db = g.sum(0)
        return (None, dw_cat[:, :c].reshape(s1), db, dw2a.reshape(s2a), d_pre.float().sum(0),
                dw_cat[:, c:].reshape(s2b), db)


# ------------------------------------------------------------------------------------------------
# Quantisers
# ------------------------------------------------------------------------------------------------
def _rows_of(z: torch.Tensor) -> torch.Tensor:
    """(b, D, h, w) -> (b*h*w, D) rows; free when z is a channels-first view of a token-major buffer."""
    return z.permute(0, 2, 3, 1).reshape(-1, z.shape[1])


def _nchw_of(rows: torch.Tensor, b: int, h: int, w: int) -> torch.Tensor:
    return rows.reshape(b, h, w, -1).permute(0, 3, 1, 2)


def _torch_normalize(z, weight, normalize):
    if normalize == "l2":
        return F.normalize(z, dim=-1), F.normalize(weight, dim=-1)
    if normalize == "z_norm":
        zs, zm = torch.std_mean(z, dim=-1, keepdim=True)
        cs, cm = torch.std_mean(weight, dim=-1, keepdim=True)
        return (z - zm) / (zs + 1e-5), (weight - cm) / (cs + 1e-5)
    if normalize == "none":
        return z, weight
    raise ValueError(f"Unsupported normalize type {normalize}")


class _PQStraightThrough(torch.autograd.Function):
    """Training-pass quantisation for l2-normalised codes with raw codeword gather (dino_pqgo.py:613-690):
        zn = z / |z| ;  emb = books[idx] ;  vq = mean_p(book * mse(emb, zn.detach()) + beta * mse(zn, emb.detach()))
        zq = zn + (emb - zn).detach()
    Forward = the eval path's HIP kernel (indices, gathered rows, squared-error sums in one pass);
    backward = the closed form in one HIP pass over the rows (``native.pq_st_backward``: z, emb and g_zq read once,
    d z written once, d books by atomic adds) instead of autograd through ~15 elementwise ops on (rows, D) tensors:
        d zn   = g_zq + g_vq * 2 beta (zn - emb) / (n d P)          d z = (d zn - zn <zn, d zn>) / |z|
        d emb  = g_vq * 2 book (emb - zn) / (n d P)                  d books[p, k] = sum over rows with idx = k
    Returns (zq rows, vq-loss, idx)."""

    @staticmethod
    def forward(ctx, rows, books, beta, book):
        num_pq, k, d = books.shape
        n = rows.shape[0]
        idx, emb, loss_sum = native.pq_assign(rows, books, "l2", False)
        vq = ((book + beta) * (loss_sum / float(n * d))).mean()
        ctx.save_for_backward(rows, emb, idx)
        ctx.meta = (num_pq, k, d, float(beta), float(book))
        ctx.mark_non_differentiable(idx)
        return emb, vq, idx

    hip_backward = True   # False: the closed form below with stock elementwise ops and index_add_ (for comparisons)

    @staticmethod
    def backward(ctx, g_zq, g_vq, _g_idx):
        rows, emb, idx = ctx.saved_tensors
        num_pq, k, d, beta, book = ctx.meta
        n = rows.shape[0]
        if _PQStraightThrough.hip_backward:
            if g_vq is None:
                g_vq = torch.zeros((), dtype=torch.float32, device=rows.device)
            g_rows, g_books = native.pq_st_backward(rows, emb, idx, g_zq, g_vq, beta, book, k, d,
                                                    want_rows=ctx.needs_input_grad[0], want_books=ctx.needs_input_grad[1])
            return g_rows, g_books, None, None
        z3 = rows.reshape(n, num_pq, d)
        norm = z3.norm(dim=-1, keepdim=True).clamp_min(1e-12)     # F.normalize's eps
        zn = z3 / norm
        diff = zn - emb.reshape(n, num_pq, d)
        scale = g_vq * (2.0 / float(n * d * num_pq))
        g_books = None
        if ctx.needs_input_grad[1]:
            flat = (idx + torch.arange(num_pq, device=idx.device).unsqueeze(1) * k).t().reshape(-1)   # (n * P,)
            g_books = torch.zeros(num_pq * k, d, dtype=rows.dtype, device=rows.device)
            g_books.index_add_(0, flat, (diff * (-book * scale)).reshape(n * num_pq, d))
            g_books = g_books.reshape(num_pq, k, d)
        g_rows = None
        if ctx.needs_input_grad[0]:
            g_zn = diff.mul_(beta * scale)
            if g_zq is not None:
                g_zn += g_zq.reshape(n, num_pq, d)
            proj = (zn * g_zn).sum(dim=-1, keepdim=True)
            g_rows = (g_zn.addcmul_(zn, proj, value=-1.0) / norm).reshape(n, num_pq * d)
        return g_rows, g_books, None, None


def _quantize_rows(rows, books, normalize, beta, book, training, vq_counts, gather_normalized=False,
                   use_restart=False, quantizers=None):
    """Shared core of Codebook / ProductQuantizerWrapper.
    rows (n, num_pq*d) fp32, books (num_pq, K, d).  Returns zq rows, idx (num_pq, n), output dict."""
    num_pq, k, d = books.shape
    n = rows.shape[0]
    output = {}
    if not (training and torch.is_grad_enabled() and (rows.requires_grad or books.requires_grad)):
        idx, zq, loss_sum = native.pq_assign(rows.detach(), books.detach(), normalize, gather_normalized)
        per_pq = loss_sum / float(n * d)                       # mse per sub-space
        output["vq-loss"] = ((book + beta) * per_pq).mean()    # dino_pqgo.py:685-687 then :771-772
    elif normalize == "l2" and not gather_normalized and rows.is_cuda:
        zq, vq, idx = _PQStraightThrough.apply(rows, books, beta, book)
        output["vq-loss"] = vq
    else:
        idx, _, _ = native.pq_assign(rows.detach(), books.detach(), normalize, gather_normalized,
                                     want_zq=False, want_loss=False)
        if use_restart:
            # a restart below writes the codebook parameter in place; the graph keeps a copy, not a view of it
            books = books.clone()
        z3 = rows.reshape(n, num_pq, d)
        zn, cn = _torch_normalize(z3, books, normalize)
        table = cn if gather_normalized else books
        emb = torch.gather(table, 1, idx.unsqueeze(-1).expand(num_pq, n, d)).permute(1, 0, 2)   # (n, num_pq, d)
        cb_loss = ((emb - zn.detach()) ** 2).mean(dim=(0, 2))
        commit = ((zn - emb.detach()) ** 2).mean(dim=(0, 2))
        output["vq-loss"] = (book * cb_loss + beta * commit).mean()
        zq = (zn + (emb - zn).detach()).reshape(n, num_pq * d)
    if training:
        with torch.no_grad():
            if idx.is_cuda:
                counts = native.code_usage(idx, k).to(rows.dtype)      # LDS histogram, all sub-spaces in one launch
            else:
                counts = torch.zeros(num_pq, k, dtype=rows.dtype, device=rows.device)
                counts.scatter_add_(1, idx, torch.ones_like(idx, dtype=rows.dtype))
            # one (num_pq, K) message instead of the reference's num_pq separate K-float all-reduces
            # (dino_pqgo.py:673 inside the python loop at :757)
            counts = all_reduce_tensor(counts, op="sum")
            # the running counts are updated BEFORE a restart, which zero-fills them (dino_pqgo.py:675-679,:541)
            if quantizers is not None:
                for p, q in enumerate(quantizers):
                    q.vq_count += counts[p]
            elif vq_counts is not None:
                vq_counts += counts
            used = (counts != 0).sum(dim=1).float() / float(k)
            output["codebook-usage"] = float(used.mean())
            if use_restart and quantizers is not None:
                zn_all, _ = _torch_normalize(rows.detach().reshape(n, num_pq, d), books.detach(), normalize)
                for p, q in enumerate(quantizers):
                    q.prepare_restart(counts[p], zn_all[:, p])
                    q.restart()
    return zq, idx, output


class Codebook(nn.Module):
    """One sub-quantiser (model/dino_pqgo.py:460-705): parameter ``embedding.weight`` (K, d) drawn
    U(-1/K, 1/K); ``vq_count`` is a plain attribute, not a buffer, as in the reference (:494)."""

    def __init__(self, num_codebook_vectors: int, latent_dim: int, beta=0.25, book=1.0, normalize: str = "none",
                 use_restart: bool = False, use_split: bool = False, use_weighted_sum: bool = False,
                 use_gumbel: bool = False, need_initialized: str = "kmeans", pq_dropout: float = 0.0,
                 jsd_ts: float = 1.0, num_query: int = 3, num_pos: int = 10):
        super().__init__()
        self.latent_dim = latent_dim
        self.beta, self.book = beta, book
        self.use_weighted_sum, self.use_gumbel = use_weighted_sum, use_gumbel
        self.pq_dropout = pq_dropout
        self.num_codebook_vectors = num_codebook_vectors
        self.embedding = nn.Embedding(num_codebook_vectors, latent_dim)
        self.embedding.weight.data.uniform_(-1.0 / num_codebook_vectors, 1.0 / num_codebook_vectors)
        self.vq_count = torch.zeros(num_codebook_vectors)
        self.update_indices = None
        self.update_candidates = None
        self.normalize = normalize
        if use_weighted_sum or use_gumbel:
            raise NotImplementedError("weighted-sum / gumbel quantisation is not part of the published "
                                      "EQUSS configuration and has no HIP kernel")
        if pq_dropout > 0.0:
            raise NotImplementedError("pq_dropout > 0 is not supported")
        if normalize not in ("l2", "z_norm", "none"):
            if normalize == "z_trainable":
                raise NotImplementedError("normalize='z_trainable' is not supported by the HIP assign kernel")
            raise ValueError(f"Unsupported normalize type {normalize}")
        self.use_restart, self.use_split = use_restart, use_split
        self.need_initialized = need_initialized
        self.jsd_ts = jsd_ts

    @torch.no_grad()
    def prepare_restart(self, vq_current_count: torch.Tensor, z_flat: torch.Tensor) -> None:
        """dino_pqgo.py:514-535: pick random rows of z to replace codewords that received no vote."""
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

    @torch.no_grad()
    def restart(self) -> None:
        if (self.update_indices is not None) and (self.update_candidates is not None):
            # written through the parameter itself (not .data): that bumps its version counter, which is
            # what the cached (num_pq, K, d) stack of the eval path is keyed on
            self.embedding.weight[self.update_indices] = self.update_candidates.float()
            self.vq_count.fill_(0)
            self.update_indices = None
            self.update_candidates = None

    @torch.no_grad()
    def initialize_from(self, z_flat: torch.Tensor) -> None:
        """First training forward re-initialises the codebook (dino_pqgo.py:589-609)."""
        mode = self.need_initialized
        if mode == "rand":
            self.prepare_restart(torch.zeros(self.num_codebook_vectors, dtype=torch.long, device=z_flat.device), z_flat)
            self.restart()
        elif mode == "kmeans":
            from sklearn.cluster import KMeans
            km = KMeans(init="k-means++", n_clusters=self.num_codebook_vectors, random_state=0)
            km.fit(z_flat.detach().cpu().numpy())
            self.embedding.weight.copy_(torch.from_numpy(km.cluster_centers_).float().to(z_flat.device))
        elif mode == "uni":
            nn.init.xavier_uniform_(self.embedding.weight)
        elif mode == "normal":
            nn.init.xavier_normal_(self.embedding.weight)
        self.need_initialized = "none"

    def forward(self, z: torch.Tensor, z_pos: torch.Tensor = None, return_prob: bool = True):
        """(b, d, h, w) -> (z_q (b,d,h,w), {"vq-loss"[, "codebook-usage"]}, distance_prob (b,h,w,K), idx (b,h,w)).
        z_pos is accepted for signature parity; nothing the reference returns depends on it."""
        b, d, h, w = z.shape
        if d not in _PQ_DIMS:
            raise ValueError(f"sub-space dim {d} is not supported by the HIP assign kernel {_PQ_DIMS}")
        rows = _rows_of(z)
        self.vq_count = self.vq_count.to(z.device)
        if self.need_initialized != "none" and self.training:
            self.initialize_from(rows.detach())
        books = self.embedding.weight.unsqueeze(0)
        zq, idx, output = _quantize_rows(rows, books, self.normalize, self.beta, self.book, self.training,
                                         None, use_restart=self.use_restart, quantizers=[self])
        prob = None
        if return_prob:
            prob = native.pq_prob(rows.detach(), books.detach(), self.normalize, self.jsd_ts).reshape(b, h, w, -1)
        return _nchw_of(zq, b, h, w), output, prob, idx.reshape(b, h, w)


class ProductQuantizerWrapper(nn.Module):
    """model/dino_pqgo.py:708-776: ``quantizers`` is a ModuleList of ``num_pq`` Codebooks over equal
    channel groups.  All sub-spaces are assigned by ONE kernel launch instead of a python loop."""

    def __init__(self, num_pq: int, num_codebook: int, embed_dim: int, beta: float = 0.25, book: float = 1.0,
                 normalize: Optional[str] = None, decay: float = 0.99, eps: float = 1e-5, use_restart: bool = False,
                 use_split: bool = False, use_gumbel: bool = False, use_weighted_sum: bool = False,
                 update_norm: bool = True, need_initialized: str = "none", pq_dropout: float = 0.0,
                 jsd_ts: float = 1.0, num_query: int = 3, num_pos: int = 10, quantizer_cls=Codebook) -> None:
        super().__init__()
        if embed_dim % num_pq != 0:
            raise ValueError(f"Embed dim {embed_dim} should be divisible by #PQ {num_pq}.")
        self.num_pq = num_pq
        self.pq_dim = embed_dim // num_pq
        self.quantizers = nn.ModuleList([
            quantizer_cls(num_codebook, self.pq_dim, beta=beta, book=book, normalize=normalize, use_restart=use_restart,
                          use_split=use_split, use_weighted_sum=use_weighted_sum, need_initialized=need_initialized,
                          pq_dropout=pq_dropout, jsd_ts=jsd_ts, num_query=num_query, num_pos=num_pos)
            for _ in range(self.num_pq)])
        self._books = None
        self._books_key = None

    def stacked_codebooks(self) -> torch.Tensor:
        """(num_pq, K, d) view of all sub-codebooks.  In training the stack is rebuilt (differentiable);
        in eval it is cached until a weight changes."""
        ws = [q.embedding.weight for q in self.quantizers]
        if self.training and torch.is_grad_enabled():
            return torch.stack(ws, dim=0)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if self._books is None or self._books_key != key:
            self._books = torch.stack([w.detach() for w in ws], dim=0).contiguous()
            self._books_key = key
        return self._books

    def invalidate_caches(self) -> None:
        """Drop the cached codebook stack.  Needed only after a weight was written in a way that leaves
        its version counter alone (``weight.data[...] = ...``, ``weight.data.copy_()``)."""
        self._books = None
        self._books_key = None

    def train(self, mode: bool = True):
        self.invalidate_caches()
        return super().train(mode)

    def forward(self, z: torch.Tensor, z_pos: torch.Tensor = None, it: int = -1, return_prob: bool = True):
        b, dim, h, w = z.shape
        if self.pq_dim not in _PQ_DIMS:
            raise ValueError(f"sub-space dim {self.pq_dim} is not supported by the HIP assign kernel {_PQ_DIMS}")
        rows = _rows_of(z)
        q0 = self.quantizers[0]
        if self.training:
            for i, q in enumerate(self.quantizers):
                q.vq_count = q.vq_count.to(z.device)
                if q.need_initialized != "none":
                    q.initialize_from(rows.detach()[:, i * self.pq_dim:(i + 1) * self.pq_dim])
        books = self.stacked_codebooks()
        zq, idx, outputs = _quantize_rows(rows, books, q0.normalize, q0.beta, q0.book, self.training, None,
                                          use_restart=q0.use_restart, quantizers=self.quantizers)
        z_q = _nchw_of(zq, b, h, w)
        z_split = torch.chunk(z, chunks=self.num_pq, dim=1)
        z_quantized = list(torch.chunk(z_q, chunks=self.num_pq, dim=1))
        idx_maps = idx.reshape(self.num_pq, b, h, w)
        z_index = [idx_maps[i] for i in range(self.num_pq)]
        prob = None
        if return_prob:
            prob = native.pq_prob(rows.detach(), books.detach(), q0.normalize, q0.jsd_ts).reshape(b, h, w, -1)
        return z_q, (z_split, z_quantized, z_index), outputs, prob


class DIONPQGO(nn.Module):
    """model/dino_pqgo.py:15-154."""

    def __init__(self, cfg: dict, cfg_loss: dict):
        super().__init__()
        self.cfg, self.cfg_loss = cfg, cfg_loss
        self.extractor = DinoFeaturizer(cfg["pretrained"])
        self.feat_dim = self.extractor.n_feats
        self.hidden_dim = cfg["vq"]["embed_dims"][0]
        self.is_dropout = cfg["pretrained"]["dropout"]
        self.dropout = nn.Dropout2d(p=cfg["pretrained"]["drop_prob"])
        self.cluster1 = nn.Sequential(nn.Conv2d(self.feat_dim, self.hidden_dim, (1, 1)))
        self.cluster2 = nn.Sequential(nn.Conv2d(self.feat_dim, self.feat_dim, (1, 1)), nn.ReLU(),
                                      nn.Conv2d(self.feat_dim, self.hidden_dim, (1, 1)))
        vq = cfg["vq"]
        vq_num_codebooks, vq_embed_dims = vq["num_codebooks"], vq["embed_dims"]
        assert len(vq_num_codebooks) == len(vq_embed_dims)
        self.embed_dim = vq_embed_dims[0]
        self.vq_num_codebooks = vq_num_codebooks[0]
        self.num_vq = len(vq_num_codebooks)
        self.beta, self.book, self.vq_type = vq["beta"], vq["book"], vq["vq_type"]
        self.normalize = vq.get("normalize", "none")
        self.use_weighted_sum = vq.get("use_weighted_sum", False)
        self.use_restart = vq.get("use_restart", False)
        self.use_split = vq.get("use_split", False)
        self.need_initialized = vq.get("need_initialized", False)
        self.pq_dropout = vq.get("pq_dropout", 0.0)
        self.jsd_ts = cfg_loss["jsd"].get("temperature", 1.0)
        self.num_query = cfg_loss["jsd"].get("num_query", 3)
        self.num_pos = cfg_loss["jsd"].get("num_pos", 10)
        self.n_kmeans = vq.get("n_kmeans", 1)
        kwargs = dict(beta=self.beta, book=self.book, normalize=self.normalize, use_restart=self.use_restart,
                      use_split=self.use_split, use_weighted_sum=self.use_weighted_sum,
                      need_initialized=self.need_initialized, pq_dropout=self.pq_dropout, jsd_ts=self.jsd_ts,
                      num_query=self.num_query, num_pos=self.num_pos)
        self.num_pq = vq.get("num_pq", 1)
        if isinstance(self.num_pq, int):
            self.num_pq = [self.num_pq] * self.num_vq
        if self.vq_type == "param":
            # the reference's `self.num_pq == 1` compares a list with 1 (dino_pqgo.py:84), so the
            # wrapper is always taken; the state_dict keys (vq_blocks.0.quantizers.i...) depend on it
            self.vq_blocks = nn.ModuleList([
                ProductQuantizerWrapper(self.num_pq[i], vq_num_codebooks[i], vq_embed_dims[i], **kwargs,
                                        quantizer_cls=Codebook) for i in range(self.num_vq)])
        elif self.vq_type == "ema":
            raise NotImplementedError("vq_type 'ema': the reference's EMACodebook.forward(z, i, it) cannot be called "
                                      "by its own wrapper (dino_pqgo.py:333 vs :758); only 'param' is usable")
        else:
            raise ValueError(f"Unsupported vq type {self.vq_type}.")
        self.stego_loss = STEGOLoss(cfg=self.cfg_loss["stego"])
        self._head = None
        self._head_key = None
        # eval forward: image ranges of a batch run on this many HIP streams (1 = everything on the caller's)
        self.eval_streams = 1    # image ranges of an eval batch on side streams (2 paid with the older block-tail kernels)
        self.eval_stream_min_images = 4   # per stream; smaller ranges leave the chip emptier than one launch does
        self._streams = None

    def invalidate_caches(self) -> None:
        """Drop every derived weight set (bf16 head and backbone weights, resized position embedding,
        stacked codebooks) so that the next forward rebuilds them.  The caches notice optimizer steps,
        load_state_dict and any in-place write made through the parameter itself (those bump its version
        counter); a write through ``param.data`` does not, and needs this call.  train() / eval() call it."""
        self._head = None
        self._head_key = None
        self.extractor.backbone.invalidate_caches()
        for blk in self.vq_blocks:
            blk.invalidate_caches()

    def train(self, mode: bool = True):
        self.invalidate_caches()
        return super().train(mode)

    # -- expansion head weights for the HIP GEMMs ---------------------------------------------------
    def _head_autograd(self, feat: torch.Tensor) -> torch.Tensor:
        """cluster1(f) + cluster2(f) (dino_pqgo.py:104-112,:127-128) for the training pass: the 1x1
        convolutions are per-token linears on token-major rows (the feature map is a channels-first
        view of token-major memory); see _HeadFunction."""
        b, c, h, w = feat.shape
        w1, w2a, w2b = self.cluster1[0], self.cluster2[0], self.cluster2[2]
        code = _HeadFunction.apply(_rows_of(feat), w1.weight, w1.bias, w2a.weight, w2a.bias, w2b.weight, w2b.bias)
        return _nchw_of(code, b, h, w)

    def _head_weights(self):
        ps = [self.cluster1[0].weight, self.cluster1[0].bias, self.cluster2[0].weight, self.cluster2[0].bias,
              self.cluster2[2].weight, self.cluster2[2].bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._head is None or self._head_key != key:
            c, d = self.feat_dim, self.hidden_dim
            w1, b1, w2a, b2a, w2b, b2b = [p.detach().float() for p in ps]
            w_cat = torch.cat((w1.reshape(d, c), w2b.reshape(d, c)), dim=1)
            self._head = dict(w2a=w2a.reshape(c, c).to(torch.bfloat16).contiguous(), b2a=b2a.contiguous(),
                              w_cat=w_cat.to(torch.bfloat16).contiguous(), b_cat=(b1 + b2b).contiguous())
            self._head_key = key
        return self._head

    @property
    def precision(self) -> str:
        """"bf16" (default): bf16 MFMA operands with fp32 accumulation in the ViT and the head, the
        throughput path.  "fp32": parity mode, every product and sum of the eval forward in fp32, so that
        `code` agrees with the reference's fp32 CPU run to ~1e-6 and the code indices agree outside exact
        near-ties.  The PQ assign is exact fp32 in both."""
        return self.extractor.backbone.precision

    @precision.setter
    def precision(self, value: str) -> None:
        if value not in ("bf16", "fp32"):
            raise ValueError(f"Unsupported precision {value}")
        self.extractor.backbone.precision = value

    @torch.no_grad()
    def _code_rows(self, img, want_feat=False):
        """img -> (code rows (b*n, D) fp32, feat (b*n, C) fp32 or None) through the HIP ViT and head."""
        if self.precision == "fp32":
            _bf, feat32 = self.extractor.forward_tokens(img, want_f32=True)
            ps = [self.cluster1[0].weight, self.cluster1[0].bias, self.cluster2[0].weight, self.cluster2[0].bias,
                  self.cluster2[2].weight, self.cluster2[2].bias]
            c, d = self.feat_dim, self.hidden_dim
            w1, b1, w2a, b2a, w2b, b2b = [p.detach().float().contiguous() for p in ps]
            code = native.expand_head_f32(feat32, w1.reshape(d, c), b1, w2a.reshape(c, c), b2a, w2b.reshape(d, c), b2b)
            return code, (feat32 if want_feat else None)
        b, _, h, w = img.shape
        p = self.extractor.patch_size
        n = (h // p) * (w // p)
        c = self.feat_dim
        fh = torch.empty((b * n, 2 * c), dtype=torch.bfloat16, device=img.device)
        out = self.extractor.forward_tokens(img, feat_bf16=fh[:, :c], want_f32=want_feat)
        feat32 = out[1] if want_feat else None
        hw = self._head_weights()
        code = native.expand_head(fh, hw["w2a"], hw["b2a"], hw["w_cat"], hw["b_cat"], self.hidden_dim)
        return code, feat32

    @torch.no_grad()
    def _forward_eval_streams(self, img):
        """The eval forward with the batch cut into ``eval_streams`` image ranges, each run ViT -> head ->
        PQ on its own HIP stream into row ranges of shared output buffers.  Images are independent and
        every kernel gives an image the same bits whatever the batch around it, so the result equals the
        single-stream one (the loss sums are added in a different order); what changes is that one
        range's store-bound epilogues and partly filled last rounds of workgroups overlap the other
        range's compute.  The caller's stream waits for all ranges before this returns."""
        b, _, h, w = img.shape
        p = self.extractor.patch_size
        gh, gw = h // p, w // p
        n, c, dim = gh * gw, self.feat_dim, self.hidden_dim
        dev = img.device
        pqw = self.vq_blocks[0]
        q0 = pqw.quantizers[0]
        if pqw.pq_dim not in _PQ_DIMS:
            raise ValueError(f"sub-space dim {pqw.pq_dim} is not supported by the HIP assign kernel {_PQ_DIMS}")
        books = pqw.stacked_codebooks()
        hw = self._head_weights()
        img = img.detach().to(torch.float32).contiguous()
        self.extractor.backbone._pos_for(h, w)     # cached tables and weight sets are built on the caller's stream
        self.extractor.backbone.device_handle()
        fh = torch.empty((b * n, 2 * c), dtype=torch.bfloat16, device=dev)
        code = torch.empty((b * n, dim), dtype=torch.float32, device=dev)
        zq = torch.empty_like(code)
        k = self.eval_streams
        if self._streams is None or len(self._streams) != k or self._streams[0].device != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        main = torch.cuda.current_stream(dev)
        start = torch.cuda.Event()
        start.record(main)
        idx_parts, loss_parts = [], []
        for i, s in enumerate(self._streams):
            lo, hi = b * i // k, b * (i + 1) // k
            r0, r1 = lo * n, hi * n
            s.wait_event(start)
            with torch.cuda.stream(s):
                self.extractor.forward_tokens(img[lo:hi], feat_bf16=fh[r0:r1, :c])
                native.expand_head(fh[r0:r1], hw["w2a"], hw["b2a"], hw["w_cat"], hw["b_cat"], dim, out=code[r0:r1])
                idx_i, _, loss_i = native.pq_assign(code[r0:r1], books, q0.normalize, False, zq_out=zq[r0:r1])
                done = torch.cuda.Event()
                done.record(s)
            main.wait_event(done)
            for t in (idx_i, loss_i):
                t.record_stream(main)       # allocated on the side stream, consumed on the caller's
            idx_parts.append(idx_i)
            loss_parts.append(loss_i)
        idx = torch.cat(idx_parts, dim=1)
        per_pq = torch.stack(loss_parts).sum(0) / float(b * n * pqw.pq_dim)
        outputs = {"vq-loss": ((q0.book + q0.beta) * per_pq).mean()}    # dino_pqgo.py:685-687 then :771-772
        code_map, z_q = _nchw_of(code, b, gh, gw), _nchw_of(zq, b, gh, gw)
        z_split = torch.chunk(code_map, chunks=pqw.num_pq, dim=1)
        z_quantized = list(torch.chunk(z_q, chunks=pqw.num_pq, dim=1))
        return code_map, z_q, (z_split, z_quantized, idx.reshape(pqw.num_pq, b, gh, gw)), outputs

    def forward(self, img: torch.Tensor, aug_img: torch.Tensor = None, img_pos: torch.Tensor = None,
                it: int = 0, stage: int = 0):
        b, _, h, w = img.shape
        p = self.extractor.patch_size
        gh, gw = h // p, w // p
        if not self.training:
            if (self.eval_streams > 1 and img.is_cuda and b >= self.eval_streams * self.eval_stream_min_images
                    and self.precision == "bf16"):
                return self._forward_eval_streams(img)
            code_rows, _ = self._code_rows(img)
            code = _nchw_of(code_rows, b, gh, gw)
            quantized, (z_split, z_quantized, z_index), outputs, _ = self.vq_blocks[0](
                code, None, it=it, return_prob=False)
            return code, quantized, (z_split, z_quantized, torch.stack(z_index, dim=0)), outputs

        # training: frozen features from the HIP ViT; head and losses through autograd
        # The image and its positive pair go through the frozen backbone as ONE batch of 2b images (every kernel treats
        # a row the same whatever batch it sits in, so the values are those of two passes): half the launches, fuller
        # tiles.  Dropout2d draws its channel masks for the image first, then for the pair, as the reference's two calls
        # do (dino_pqgo.py:122-131).  The head stays two calls: slices of one batched output would each bring a
        # zero-filled full-size gradient into the backward.
        if img_pos.shape == img.shape:
            both = self.extractor(torch.cat((img, img_pos), dim=0))
            if self.training and self.dropout.p > 0.0:
                F.dropout2d(both[:b], self.dropout.p, True, inplace=True)
                F.dropout2d(both[b:], self.dropout.p, True, inplace=True)
            dino_feat, dino_feat_pos = both[:b], both[b:]
            code = self._head_autograd(dino_feat)
            code_pos = self._head_autograd(dino_feat_pos)
        else:
            dino_feat = self.dropout(self.extractor(img))
            code = self._head_autograd(dino_feat)
            dino_feat_pos = self.dropout(self.extractor(img_pos))
            code_pos = self._head_autograd(dino_feat_pos)
        quantized, (z_split, z_quantized, z_index), outputs, _ = self.vq_blocks[0](
            code, code_pos, it=it, return_prob=False)
        start = time.time()
        outputs["stego-loss"] = self.stego_loss(dino_feat, dino_feat_pos, code, code_pos)
        # the reference brackets this with two host synchronisations (dino_pqgo.py:145-148) to time it;
        # the key is kept for the logging code, the syncs are not
        outputs["end_time"] = time.time() - start
        return code, quantized, (z_split, z_quantized, torch.stack(z_index, dim=0)), outputs
