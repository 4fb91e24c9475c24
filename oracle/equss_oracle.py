"""CPU oracle for the EQUSS hot path.  TEST INFRASTRUCTURE ONLY.

A functional fp32 restatement, on stock PyTorch CPU ops, of what the reference computes for
DINO-ViT patch features -> expansion head -> product-quantisation assign (and the probes next to
it).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the shipped package never does.

Pinning: oracle/gen_golden.py imports the reference itself (CPU, fp32) in the build container and
asserts that every function here reproduces the reference's outputs bit for bit on the same
seeded weights and inputs, then freezes those outputs as fixtures in tests/golden/.  The reference
has no tests or golden vectors of its own, so those fixtures are what pins parity.

All state is passed as a plain ``dict[str, Tensor]`` with the reference's state_dict key names.
Each function cites the reference lines it follows.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

__all__ = [
    "vit_interpolate_pos_encoding", "vit_prepare_tokens", "vit_attention", "vit_block",
    "vit_last_feat", "dino_featurizer", "expansion_head", "codebook_forward", "pq_forward",
    "pqgo_forward", "cluster_lookup", "evaluator_forward", "wrapper_forward",
    "vector_quantizer_forward", "ema_vector_quantizer_eval", "histogram_count", "pq_top2_gap",
    "evaluator_top2_gap", "metrics_confusion", "metrics_compute",
    "stego_sample", "stego_loss", "train_forward", "TRAINABLE_PREFIXES",
    "quantizer_pq_eval", "ema_vector_quantizer_train",
]

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# DINO ViT  (model/dino/vision_transformer.py)
# --------------------------------------------------------------------------------------------
def vit_interpolate_pos_encoding(pos_embed: torch.Tensor, npatch: int, w: int, h: int,
                                 patch_size: int) -> torch.Tensor:
    """vision_transformer.py:178-198.  ``w``/``h`` are the image's dims 2 and 3 as the reference
    names them; bicubic resize of the (sqrt(N) x sqrt(N)) grid with the +0.1 scale trick."""
    n = pos_embed.shape[1] - 1
    if npatch == n and w == h:
        return pos_embed
    class_pos = pos_embed[:, 0]
    patch_pos = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0 = w // patch_size + 0.1
    h0 = h // patch_size + 0.1
    side = int(math.sqrt(n))
    patch_pos = F.interpolate(
        patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2),
        scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)),
        mode="bicubic",
    )
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).view(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)


def vit_prepare_tokens(sd: State, img: torch.Tensor, patch_size: int, prefix: str = "") -> torch.Tensor:
    """vision_transformer.py:130-133 (patch conv, flatten, transpose) and :200-211."""
    b, _, w, h = img.shape
    x = F.conv2d(img, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=patch_size)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[prefix + "cls_token"].expand(b, -1, -1)
    x = torch.cat((cls, x), dim=1)
    x = x + vit_interpolate_pos_encoding(sd[prefix + "pos_embed"], x.shape[1] - 1, w, h, patch_size)
    return x


def vit_attention(sd: State, x: torch.Tensor, num_heads: int, prefix: str) -> torch.Tensor:
    """vision_transformer.py:79-91 (the attention matrix and qkv it also returns are dropped by
    the caller, dino_featurizer.py:76-80)."""
    b, n, c = x.shape
    hd = c // num_heads
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd[prefix + "qkv.bias"])
    qkv = qkv.reshape(b, n, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def vit_block(sd: State, x: torch.Tensor, num_heads: int, prefix: str) -> torch.Tensor:
    """vision_transformer.py:106-114 with LayerNorm eps 1e-6 (:271) and exact-erf GELU (:49,:58-64)."""
    c = x.shape[-1]
    y = F.layer_norm(x, (c,), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], 1e-6)
    x = x + vit_attention(sd, y, num_heads, prefix + "attn.")
    y = F.layer_norm(x, (c,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 1e-6)
    y = F.linear(y, sd[prefix + "mlp.fc1.weight"], sd[prefix + "mlp.fc1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[prefix + "mlp.fc2.weight"], sd[prefix + "mlp.fc2.bias"])
    return x + y


def vit_last_feat(sd: State, img: torch.Tensor, patch_size: int, num_heads: int, depth: int,
                  prefix: str = "", taps: Optional[List[int]] = None):
    """vision_transformer.py:227-239 with n=1: all blocks, then the final norm on the last
    block's output.  ``taps`` optionally returns the residual stream after the listed blocks."""
    x = vit_prepare_tokens(sd, img, patch_size, prefix)
    tapped = {}
    if taps is not None and -1 in taps:
        tapped[-1] = x
    for i in range(depth):
        x = vit_block(sd, x, num_heads, f"{prefix}blocks.{i}.")
        if taps is not None and i in taps:
            tapped[i] = x
    c = x.shape[-1]
    feat = F.layer_norm(x, (c,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6)
    return (feat, tapped) if taps is not None else feat


def dino_featurizer(sd: State, img: torch.Tensor, patch_size: int, num_heads: int, depth: int,
                    prefix: str = "") -> torch.Tensor:
    """dino_featurizer.py:64-83: drop CLS, (b, n, C) -> (b, C, h/p, w/p) contiguous."""
    b, _, h, w = img.shape
    assert h % patch_size == 0 and w % patch_size == 0
    fh, fw = h // patch_size, w // patch_size
    feat = vit_last_feat(sd, img, patch_size, num_heads, depth, prefix)
    return feat[:, 1:, :].reshape(b, fh, fw, -1).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------------
# Expansion head  (model/dino_pqgo.py:104-112, applied :126-128; eval: Dropout2d is identity)
# --------------------------------------------------------------------------------------------
def expansion_head(sd: State, feat: torch.Tensor, prefix: str = "") -> torch.Tensor:
    code = F.conv2d(feat, sd[prefix + "cluster1.0.weight"], sd[prefix + "cluster1.0.bias"])
    y = F.conv2d(feat, sd[prefix + "cluster2.0.weight"], sd[prefix + "cluster2.0.bias"])
    y = F.relu(y)
    y = F.conv2d(y, sd[prefix + "cluster2.2.weight"], sd[prefix + "cluster2.2.bias"])
    code += y
    return code


# --------------------------------------------------------------------------------------------
# Product quantisation  (model/dino_pqgo.py:460-776)
# --------------------------------------------------------------------------------------------
def _normalize_pair(z_flat, codebook, normalize):
    """dino_pqgo.py:613-640 (l2 / z_norm / none; z_trainable needs module parameters and is
    covered by the module-level tests only)."""
    if normalize == "l2":
        return F.normalize(z_flat, dim=1), F.normalize(codebook, dim=1)
    if normalize == "z_norm":
        z_std, z_mean = torch.std_mean(z_flat, dim=1, keepdim=True)
        c_std, c_mean = torch.std_mean(codebook, dim=1, keepdim=True)
        return (z_flat - z_mean) / (z_std + 1e-5), (codebook - c_mean) / (c_std + 1e-5)
    if normalize == "none":
        return z_flat, codebook
    raise ValueError(f"Unsupported normalize type {normalize}")


def _sq_distance(z_norm, codebook_norm):
    """dino_pqgo.py:646-648 / quantizer.py:140-144: (sum z^2 + sum c^2) - 2 z c^T, in that order."""
    return torch.sum(z_norm ** 2, dim=1, keepdim=True) + \
        torch.sum(codebook_norm ** 2, dim=1) - \
        2 * (torch.matmul(z_norm, codebook_norm.t()))


def codebook_forward(weight: torch.Tensor, z: torch.Tensor, normalize: str = "l2", beta: float = 0.25,
                     book: float = 1.0, jsd_ts: float = 1.0, training: bool = False):
    """Codebook.forward, dino_pqgo.py:579-705, for the published configuration
    (use_weighted_sum = use_gumbel = use_restart = False, pq_dropout = 0) after any first-call
    re-initialisation.  Returns (z_q (b,d,h,w), output dict, distance_prob (b,h,w,K), idx (b,h,w));
    in training mode the dict also holds ``codebook-usage`` and the per-code counts are returned
    under ``_count`` (single process: the all-reduce at :673 is the identity)."""
    output = {}
    b, d, h, w = z.shape
    zp = z.permute(0, 2, 3, 1).contiguous()
    z_flat = zp.view(-1, d)
    z_norm, codebook_norm = _normalize_pair(z_flat, weight, normalize)
    dist = _sq_distance(z_norm, codebook_norm)
    idx = torch.argmin(dist, dim=1)
    prob = F.softmax(-dist / jsd_ts, dim=1)
    z_q = F.embedding(idx, weight)  # raw rows, dino_pqgo.py:665
    if training:
        count = torch.sum(F.one_hot(idx, weight.shape[0]).to(z.dtype), dim=0)
        output["codebook-usage"] = (codebook_norm.shape[0] - len(
            torch.nonzero(count == 0, as_tuple=True)[0])) / codebook_norm.shape[0]
        output["_count"] = count
    codebook_loss = F.mse_loss(z_q, z_norm.detach())
    commitment_loss = F.mse_loss(z_norm, z_q.detach())
    q_loss = book * codebook_loss + beta * commitment_loss
    z_q = z_norm + (z_q - z_norm).detach()
    z_q = z_q.view(zp.shape).permute(0, 3, 1, 2).contiguous()
    output["vq-loss"] = q_loss
    return z_q, output, prob.view(b, h, w, -1).contiguous(), idx.view(b, h, w).contiguous()


def pq_forward(weights: List[torch.Tensor], z: torch.Tensor, normalize: str = "l2", beta: float = 0.25,
               book: float = 1.0, jsd_ts: float = 1.0, training: bool = False):
    """ProductQuantizerWrapper.forward, dino_pqgo.py:745-776 (the z_pos half is computed by the
    reference but never consumed in eval; it does not influence any returned value)."""
    num_pq = len(weights)
    z_split = torch.chunk(z, chunks=num_pq, dim=1)
    z_quantized, z_index, probs = [], [], []
    outputs = {}
    for i in range(num_pq):
        q_i, out_i, prob_i, idx_i = codebook_forward(weights[i], z_split[i], normalize, beta, book,
                                                     jsd_ts, training)
        z_quantized.append(q_i)
        z_index.append(idx_i)
        for k, v in out_i.items():
            if k.startswith("_"):
                outputs.setdefault(k, []).append(v)
            else:
                outputs[k] = v if i == 0 else outputs[k] + v
        probs.append(prob_i)
    for k in list(outputs):
        if not k.startswith("_"):
            outputs[k] = outputs[k] / num_pq
    return torch.cat(z_quantized, dim=1), (z_split, z_quantized, z_index), outputs, torch.cat(probs, dim=-1)


def _codebooks(sd: State, prefix: str) -> List[torch.Tensor]:
    single = prefix + "vq_blocks.0.embedding.weight"
    if single in sd:
        return [sd[single]]
    out, i = [], 0
    while f"{prefix}vq_blocks.0.quantizers.{i}.embedding.weight" in sd:
        out.append(sd[f"{prefix}vq_blocks.0.quantizers.{i}.embedding.weight"])
        i += 1
    return out


def pqgo_forward(sd: State, img: torch.Tensor, cfg: dict, prefix: str = "model."):
    """DIONPQGO.forward in eval mode, dino_pqgo.py:114-154.
    Returns (code, quantized_feat, stacked indices (num_pq,b,h,w), outputs)."""
    from_cfg = cfg["model"]
    arch = {"vit_tiny": (3, 12), "vit_small": (6, 12), "vit_base": (12, 12)}[from_cfg["pretrained"]["model_type"]]
    patch = from_cfg["pretrained"]["dino_patch_size"]
    feat = dino_featurizer(sd, img, patch, arch[0], arch[1], prefix + "extractor.backbone.")
    code = expansion_head(sd, feat, prefix)
    vq = from_cfg["vq"]
    quantized, (_zs, _zq, z_index), outputs, _prob = pq_forward(
        _codebooks(sd, prefix), code, vq.get("normalize", "none"), vq["beta"], vq["book"],
        cfg["loss"]["jsd"].get("temperature", 1.0), False)
    return code, quantized, torch.stack(z_index, dim=0), outputs


@torch.no_grad()
def pq_top2_gap(weights: List[torch.Tensor], code_rows: torch.Tensor, normalize: str = "l2"):
    """fp64 nearest / second-nearest squared distances per (sub-space, row) for rows of a
    token-major ``code`` (n, D): returns (idx (num_pq,n) int64, gap (num_pq,n) float64).  Used by
    the tests to tell genuine mismatches from near-ties an fp32 sum may legitimately flip."""
    num_pq = len(weights)
    d = code_rows.shape[1] // num_pq
    idxs, gaps = [], []
    for i in range(num_pq):
        z = code_rows[:, i * d:(i + 1) * d].double()
        zn, cn = _normalize_pair(z, weights[i].double(), normalize)
        dist = _sq_distance(zn, cn)
        top2 = torch.topk(dist, 2, dim=1, largest=False)
        idxs.append(top2.indices[:, 0])
        gaps.append(top2.values[:, 1] - top2.values[:, 0])
    return torch.stack(idxs), torch.stack(gaps)


# --------------------------------------------------------------------------------------------
# Training step  (model/loss.py:647-739, model/dino_pqgo.py:114-154 with self.training,
# wrapper/PQGOWrapper.py:52-103)
# --------------------------------------------------------------------------------------------
def stego_sample(t: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """loss.py:655-656."""
    return F.grid_sample(t, coords.permute(0, 2, 1, 3), padding_mode="border", align_corners=True)


def _stego_helper(cfg, f1, f2, c1, c2, shift):
    """STEGOLoss.helper, loss.py:677-699."""
    unit = lambda t: F.normalize(t, dim=1, eps=1e-10)   # noqa: E731  (loss.py:651-652)
    with torch.no_grad():
        fd = torch.einsum("nchw,ncij->nhwij", unit(f1), unit(f2))
        if cfg["pointwise"]:
            old_mean = fd.mean()
            fd -= fd.mean([3, 4], keepdim=True)
            fd = fd - fd.mean() + old_mean
    cd = torch.einsum("nchw,ncij->nhwij", unit(c1), unit(c2))
    min_val = 0.0 if cfg["zero_clamp"] else -9999.0
    if cfg["stabilize"]:
        return -cd.clamp(min_val, .8) * (fd - shift)
    return -cd.clamp(min_val) * (fd - shift)


def stego_loss(cfg: dict, feats, feats_pos, code, code_pos, coords1, coords2, perms):
    """STEGOLoss.forward, loss.py:701-739, with its random draws handed in: coords1 / coords2
    (b, n, n, 2) in [-1, 1) are the two torch.rand draws of :709-710 (already scaled), perms the
    ``neg_samples`` outputs of super_perm (:659-663, :725) in call order."""
    f = stego_sample(feats, coords1)
    c = stego_sample(code, coords1)
    f_pos = stego_sample(feats_pos, coords2)
    c_pos = stego_sample(code_pos, coords2)
    pos_intra = _stego_helper(cfg, f, f, c, c, cfg["pos_intra_shift"])
    pos_inter = _stego_helper(cfg, f, f_pos, c, c_pos, cfg["pos_inter_shift"])
    negs = []
    for perm in perms:
        f_neg = stego_sample(feats[perm], coords2)
        c_neg = stego_sample(code[perm], coords2)
        negs.append(_stego_helper(cfg, f, f_neg, c, c_neg, cfg["neg_inter_shift"]))
    neg = torch.cat(negs, dim=0)
    return (cfg["pos_intra_weight"] * pos_intra.mean() + cfg["pos_inter_weight"] * pos_inter.mean() +
            cfg["neg_inter_weight"] * neg.mean())


# the parameters train.py optimises (head, codebooks, probes); the backbone is frozen
TRAINABLE_PREFIXES = ("model.cluster1.", "model.cluster2.", "model.vq_blocks.", "evaluator.")


def train_forward(sd: State, img, img_pos, label, cfg: dict, coords1, coords2, perms, backward: bool = True):
    """One training forward (+ backward) of PQGOWrapper(DIONPQGO) in fp32 on the CPU:
    dino_pqgo.py:125-150 with self.training (second ViT pass on img_pos, head on both, PQ, STEGO
    loss), PQGOWrapper.py:71-101 (loss weights, probes on the detached quantised map).  Restated for
    need_initialized = "none", use_restart = False and drop_prob = 0 (Dropout2d is then the identity,
    dino_pqgo.py:126,:133); the STEGO draws are arguments.  Returns (total_loss (1,), output dict,
    {key: gradient} for every trainable tensor when ``backward``)."""
    m = cfg["model"]
    assert m["pretrained"]["drop_prob"] == 0.0 and m["vq"].get("need_initialized", "none") == "none"
    heads, depth = {"vit_tiny": (3, 12), "vit_small": (6, 12), "vit_base": (12, 12)}[m["pretrained"]["model_type"]]
    patch = m["pretrained"]["dino_patch_size"]
    p = {k: (v.detach().clone().requires_grad_(True) if k.startswith(TRAINABLE_PREFIXES) else v) for k, v in sd.items()}
    with torch.enable_grad():
        with torch.no_grad():    # frozen backbone, dino_featurizer.py:72-75
            feat = dino_featurizer(p, img, patch, heads, depth, "model.extractor.backbone.")
            feat_pos = dino_featurizer(p, img_pos, patch, heads, depth, "model.extractor.backbone.")
        code = expansion_head(p, feat, "model.")
        code_pos = expansion_head(p, feat_pos, "model.")
        vq = m["vq"]
        quantized, (_zs, _zq, z_index), outputs, _prob = pq_forward(
            _codebooks(p, "model."), code, vq.get("normalize", "none"), vq["beta"], vq["book"],
            cfg["loss"]["jsd"].get("temperature", 1.0), True)
        counts = outputs.pop("_count")
        outputs["stego-loss"] = stego_loss(cfg["loss"]["stego"], feat, feat_pos, code, code_pos, coords1, coords2, perms)
        model_loss = torch.zeros(1)
        if cfg["loss"]["stego_weight"] > 0.0:
            model_loss = model_loss + outputs["stego-loss"] * cfg["loss"]["stego_weight"]
        if cfg["loss"]["vq_weight"] > 0.0:
            model_loss = model_loss + outputs["vq-loss"] * cfg["loss"]["vq_weight"]
        outputs["loss"] = model_loss
        otype = cfg["eval"]["output_type"]
        out = torch.clone(code if otype == "feat" else quantized).detach()
        linear_loss, _lp, cluster_loss, _cp = evaluator_forward(p, out, label, cfg["num_classes"])
        outputs["linear-loss"] = linear_loss
        outputs["cluster-loss"] = cluster_loss
        total = model_loss + linear_loss + cluster_loss
        grads = {}
        if backward:
            total.backward()
            grads = {k: v.grad for k, v in p.items() if k.startswith(TRAINABLE_PREFIXES)}
    outputs["_count"] = torch.stack(counts)
    outputs["_index"] = torch.stack(z_index, dim=0)
    return total.detach(), outputs, grads


# --------------------------------------------------------------------------------------------
# Probes  (model/evaluator.py)
# --------------------------------------------------------------------------------------------
def cluster_lookup(clusters: torch.Tensor, x: torch.Tensor, alpha: Optional[float] = None):
    """ClusterLookup.forward, evaluator.py:93-111 (alpha None -> one-hot of the arg-max)."""
    normed_clusters = F.normalize(clusters, dim=1)
    normed_features = F.normalize(x, dim=1)
    inner = torch.einsum("bchw,nc->bnhw", normed_features, normed_clusters)
    if alpha is None:
        probs = F.one_hot(torch.argmax(inner, dim=1), clusters.shape[0])
        probs = probs.permute(0, 3, 1, 2).contiguous().to(torch.float32)
    else:
        probs = F.softmax(inner * alpha, dim=1)
    loss = -torch.sum(probs * inner, dim=1).mean()
    return loss, probs


def evaluator_forward(sd: State, out: torch.Tensor, label: torch.Tensor, num_classes: int,
                      prefix: str = "evaluator."):
    """UnSegEvaluator.forward with is_crf=False, evaluator.py:46-82."""
    if out.shape[-2:] != label.shape[-2:]:
        out = F.interpolate(out, label.shape[-2:], mode="bilinear", align_corners=False)
    linear_logits = F.conv2d(out, sd[prefix + "linear_probe.weight"], sd[prefix + "linear_probe.bias"])
    linear_preds = linear_logits.argmax(1)
    cluster_loss, cluster_probs = cluster_lookup(sd[prefix + "cluster_probe.clusters"], out, None)
    cluster_preds = cluster_probs.argmax(1)
    label_flat = label.reshape(-1)
    mask = torch.logical_and(label_flat >= 0, label_flat < num_classes)
    linear_logits = F.interpolate(linear_logits, label.shape[-2:], mode="bilinear", align_corners=False)
    logit_flat = linear_logits.permute(0, 2, 3, 1).reshape(-1, num_classes)
    linear_loss = F.cross_entropy(logit_flat[mask], label_flat[mask]).mean()
    return linear_loss, linear_preds, cluster_loss, cluster_preds


def wrapper_forward(sd: State, img: torch.Tensor, label: torch.Tensor, cfg: dict):
    """PQGOWrapper.forward in eval mode, wrapper/PQGOWrapper.py:52-103."""
    code, feat_vqs, z_index, output = pqgo_forward(sd, img, cfg, "model.")
    model_loss = torch.zeros(1)
    if cfg["loss"]["vq_weight"] > 0.0:
        model_loss = model_loss + output["vq-loss"] * cfg["loss"]["vq_weight"]
    output = dict(output)
    output["loss"] = model_loss
    otype = cfg["eval"]["output_type"]
    if otype == "feat":
        out = torch.clone(code).detach()
    elif otype[:2] == "vq":
        out = torch.clone(feat_vqs).detach()
    else:
        raise ValueError(f"Unsupported output type {otype}.")
    linear_loss, linear_preds, cluster_loss, cluster_preds = evaluator_forward(
        sd, out, label, cfg["num_classes"])
    output["linear-loss"] = linear_loss
    output["cluster-loss"] = cluster_loss
    total = model_loss + linear_loss + cluster_loss
    return total, output, (linear_preds, cluster_preds), feat_vqs, z_index


def evaluator_top2_gap(sd: State, out: torch.Tensor, label_hw, prefix: str = "evaluator."):
    """fp64 evaluation of both probes (evaluator.py:53-70, :95-98): arg-max and the gap between the
    largest and second-largest class score per output pixel.  Tests use the gap to tell a genuine
    mismatch from a near-tie that fp32 rounding order may resolve either way."""
    o = out.double()
    if tuple(o.shape[-2:]) != tuple(label_hw):
        o = F.interpolate(o, tuple(label_hw), mode="bilinear", align_corners=False)
    logits = F.conv2d(o, sd[prefix + "linear_probe.weight"].double(), sd[prefix + "linear_probe.bias"].double())
    inner = torch.einsum("bchw,nc->bnhw", F.normalize(o, dim=1),
                         F.normalize(sd[prefix + "cluster_probe.clusters"].double(), dim=1))
    res = []
    for score in (logits, inner):
        top = score.topk(2, dim=1)
        res.append((top.indices[:, 0], top.values[:, 0] - top.values[:, 1]))
    return res[0], res[1]


def metrics_confusion(preds: torch.Tensor, label: torch.Tensor, num_classes: int, extra_classes: int = 0):
    """UnSegMetrics.update, metric.py:45-58: confusion[pred, label] over pixels whose label AND
    prediction lie in [0, num_classes)."""
    preds, label = preds.reshape(-1), label.reshape(-1)
    n_pred = num_classes + extra_classes
    mask = (label >= 0) & (label < num_classes) & (preds >= 0) & (preds < num_classes)
    conf = torch.zeros(n_pred, num_classes, dtype=torch.long)
    for p, l in zip(preds[mask].tolist(), label[mask].tolist()):
        conf[p, l] += 1
    return conf


def metrics_compute(confusion: torch.Tensor, compute_hungarian: bool, num_classes: int, extra_classes: int = 0):
    """UnSegMetrics.compute, metric.py:60-98, on one rank: (iou %, accuracy %, histogram, assignments)."""
    from scipy.optimize import linear_sum_assignment
    import numpy as np
    if compute_hungarian:
        assignments = linear_sum_assignment(confusion, maximize=True)
        if extra_classes == 0:
            histogram = confusion[np.argsort(assignments[1]), :]
        else:
            assignments_t = linear_sum_assignment(confusion.t(), maximize=True)
            histogram = confusion[assignments_t[1], :]
            missing = list(set(range(num_classes + extra_classes)) - set(assignments[0]))
            histogram = torch.cat([histogram, confusion[missing, :].sum(0, keepdim=True)], dim=0)
            histogram = torch.cat([histogram, torch.zeros(num_classes + 1, 1)], dim=1)
    else:
        assignments = (torch.arange(num_classes).unsqueeze(1), torch.arange(num_classes).unsqueeze(1))
        histogram = confusion
    tp = torch.diag(histogram)
    fp = torch.sum(histogram, dim=0) - tp
    fn = torch.sum(histogram, dim=1) - tp
    iou = tp / (tp + fp + fn)
    iou = iou[~torch.isnan(iou)].mean()
    accuracy = torch.sum(tp) / torch.sum(histogram)
    return 100 * iou, 100 * accuracy, histogram, assignments


# --------------------------------------------------------------------------------------------
# model/quantizer.py family (flat or NCHW input, normalised rows gathered)
# --------------------------------------------------------------------------------------------
def histogram_count(count: torch.Tensor, prefix: str = "") -> dict:
    """get_histogram_count, quantizer.py:16-30: first sorted-rank fraction at which the cumulative
    usage reaches 10 / 50 / 90 %."""
    prob = count.float() / (count.sum() + 1)
    k = len(prob)
    prob, _ = torch.sort(prob, dim=0, descending=True)
    c_sum = torch.cumsum(prob, dim=0)
    out = {}
    for name, thr in (("p10", 0.1), ("p50", 0.5), ("p90", 0.9)):
        hit = torch.nonzero(c_sum >= thr, as_tuple=True)[0]
        out[f"{prefix}-{name}"] = (int(hit[0]) / k) if len(hit) else None
    return out


def vector_quantizer_forward(weight: torch.Tensor, z: torch.Tensor, normalize: str = "l2",
                             beta: float = 0.25):
    """VectorQuantizer.forward, quantizer.py:105-189, without the running-count bookkeeping:
    returns (q (b,d,h,w), {loss, codebook_loss, commitment_loss}, distance_prob (n,K), idx (n,))."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    b, h, w, d = zp.shape
    z_flat = zp.view(-1, d)
    z_norm, codebook_norm = _normalize_pair(z_flat, weight, normalize)
    dist = _sq_distance(z_norm, codebook_norm)
    idx = torch.argmin(dist, dim=1)
    prob = F.softmax(-dist, dim=1)
    zq = F.embedding(idx, codebook_norm)
    codebook_loss = F.mse_loss(zq, z_norm.detach())
    commitment_loss = F.mse_loss(z_norm, zq.detach())
    out = {"loss": codebook_loss + beta * commitment_loss, "codebook_loss": codebook_loss,
           "commitment_loss": commitment_loss}
    zq = z_norm + (zq - z_norm).detach()
    return zq.view(b, h, w, d).permute(0, 3, 1, 2).contiguous(), out, prob, idx


def ema_vector_quantizer_eval(weight: torch.Tensor, z_flat: torch.Tensor, normalize: str = "l2",
                              beta: float = 0.25, update_norm: bool = True):
    """EMAVectorQuantizer.forward in eval mode on flat (n,d) input, quantizer.py:383-542."""
    z_norm, codebook_norm = _normalize_pair(z_flat, weight, normalize)
    dist = _sq_distance(z_norm, codebook_norm)
    idx = torch.argmin(dist, dim=1)
    prob = F.softmax(-dist * 1.0, dim=1)
    zq = F.embedding(idx, codebook_norm) if update_norm else F.embedding(idx, weight)
    commitment_loss = F.mse_loss(z_norm, zq.detach())
    out = {"loss": beta * commitment_loss, "commitment-loss": commitment_loss,
           "codebook-sum": torch.sum(torch.abs(weight))}
    zq = z_norm + (zq - z_norm).detach()
    return zq, out, prob, idx


def quantizer_pq_eval(weights: List[torch.Tensor], z_flat: torch.Tensor, normalize: str = "l2", beta: float = 0.25,
                      update_norm: bool = True):
    """model/quantizer.py's ProductQuantizerWrapper.forward (:587-611) over EMAVectorQuantizers in eval
    mode on flat (n, D) input: chunk the channels, quantise each group, concatenate, average the dicts."""
    num_pq = len(weights)
    qs, probs, outputs = [], [], {}
    for i, z_i in enumerate(torch.chunk(z_flat, chunks=num_pq, dim=1)):
        q_i, out_i, prob_i, _idx = ema_vector_quantizer_eval(weights[i], z_i, normalize, beta, update_norm)
        qs.append(q_i)
        probs.append(prob_i)
        for k, v in out_i.items():
            outputs[k] = v if i == 0 else outputs[k] + v
    for k in outputs:
        outputs[k] = outputs[k] / num_pq
    return torch.cat(qs, dim=1), outputs, torch.cat(probs, dim=-1)


def ema_vector_quantizer_train(state: Dict[str, torch.Tensor], z_flat: torch.Tensor, normalize: str = "l2",
                               beta: float = 0.25, decay: float = 0.99, eps: float = 1e-5, update_norm: bool = True):
    """EMAVectorQuantizer.forward in TRAINING mode (quantizer.py:383-542 with :480-510) on one process,
    after any first-call initialisation, use_restart = use_split = False.  ``state`` holds the buffers
    ``weight``, ``weight_avg``, ``ema_count`` (EmbeddingEMA.vq_count, :221) and ``vq_count`` (the module's
    own running count, :493) and is UPDATED IN PLACE as the reference updates its buffers (:233-254).
    Returns (z_q (n, d), output dict, distance_prob (n, K), idx (n,))."""
    weight = state["weight"]
    k = weight.shape[0]
    z_norm, codebook_norm = _normalize_pair(z_flat, weight, normalize)
    dist = _sq_distance(z_norm, codebook_norm)
    idx = torch.argmin(dist, dim=1)
    prob = F.softmax(-dist * 1.0, dim=1)
    zq = F.embedding(idx, codebook_norm) if update_norm else F.embedding(idx, weight)
    out = {}
    with torch.no_grad():
        one_hot = F.one_hot(idx, k).to(z_flat.dtype)
        cur = torch.sum(one_hot, dim=0)
        cur_sum = torch.matmul(one_hot.t(), z_flat)
        state["vq_count"] += cur
        out.update(histogram_count(state["vq_count"], prefix="total"))
        out.update(histogram_count(cur, prefix="current"))
        state["ema_count"].mul_(decay).add_(cur, alpha=1 - decay)
        state["weight_avg"].mul_(decay).add_(cur_sum, alpha=1 - decay)
        n = state["ema_count"].sum()
        smoothed = (state["ema_count"] + eps) / (n + k * eps) * n
        new_weight = state["weight_avg"] / smoothed.unsqueeze(1)
        out["codebook-usage"] = (k - len(torch.nonzero(cur == 0, as_tuple=True)[0])) / k
    commitment_loss = F.mse_loss(z_norm, zq.detach())
    state["weight"] = new_weight          # the reference copies into the buffer before reading it at :532
    out["loss"] = beta * commitment_loss
    out["commitment-loss"] = commitment_loss
    out["codebook-sum"] = torch.sum(torch.abs(new_weight))
    zq = z_norm + (zq - z_norm).detach()
    return zq, out, prob, idx
