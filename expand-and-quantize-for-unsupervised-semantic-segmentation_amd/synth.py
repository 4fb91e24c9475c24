"""Synthetic weights and inputs, reproducible from a seed on any machine.

There is no network for the DINO checkpoints (model/dino/dino_featurizer.py:48-51 fetches them)
and no dataset, so every test, fixture and benchmark regenerates weights and images from
numpy's PCG64 stream, which is stable across numpy versions and platforms.  Tensors are drawn in
sorted key order so that the reference import, the CPU oracle and the HIP path all see the same
bits.  Scales are chosen so activations look like a trained network's: unit-variance q/k (softmax
logits with std ~1), non-zero biases, non-trivial LayerNorm gains.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from .configs import VIT_ARCH

__all__ = ["synth_backbone_state", "synth_pqgo_state", "synth_images", "synth_labels",
           "wrapper_state_shapes"]


def _backbone_shapes(model_type, patch_size, pos_grid=28):
    c, depth, _heads = VIT_ARCH[model_type]
    shapes = OrderedDict()
    shapes["cls_token"] = (1, 1, c)
    shapes["pos_embed"] = (1, pos_grid * pos_grid + 1, c)  # img_size 224 / patch 8 (vision_transformer.py:146-151)
    shapes["patch_embed.proj.weight"] = (c, 3, patch_size, patch_size)
    shapes["patch_embed.proj.bias"] = (c,)
    for i in range(depth):
        p = f"blocks.{i}."
        shapes[p + "norm1.weight"] = (c,)
        shapes[p + "norm1.bias"] = (c,)
        shapes[p + "attn.qkv.weight"] = (3 * c, c)
        shapes[p + "attn.qkv.bias"] = (3 * c,)
        shapes[p + "attn.proj.weight"] = (c, c)
        shapes[p + "attn.proj.bias"] = (c,)
        shapes[p + "norm2.weight"] = (c,)
        shapes[p + "norm2.bias"] = (c,)
        shapes[p + "mlp.fc1.weight"] = (4 * c, c)
        shapes[p + "mlp.fc1.bias"] = (4 * c,)
        shapes[p + "mlp.fc2.weight"] = (c, 4 * c)
        shapes[p + "mlp.fc2.bias"] = (c,)
    shapes["norm.weight"] = (c,)
    shapes["norm.bias"] = (c,)
    return shapes


def wrapper_state_shapes(cfg):
    """Key -> shape for the whole PQGOWrapper state_dict (wrapper/PQGOWrapper.py:21-50)."""
    pre = cfg["model"]["pretrained"]
    vq = cfg["model"]["vq"]
    c = VIT_ARCH[pre["model_type"]][0]
    d_out = vq["embed_dims"][0]
    k = vq["num_codebooks"][0]
    num_pq = vq["num_pq"][0] if isinstance(vq["num_pq"], (list, tuple)) else vq["num_pq"]
    n_cls = cfg["num_classes"]
    extra = cfg["eval"]["extra_classes"]
    # pos_embed grid: the DINO checkpoints are trained at 224 px (vision_transformer.py:139,146)
    grid = 224 // pre["dino_patch_size"]
    shapes = OrderedDict()
    for key, shp in _backbone_shapes(pre["model_type"], pre["dino_patch_size"], grid).items():
        shapes["model.extractor.backbone." + key] = shp
    shapes["model.cluster1.0.weight"] = (d_out, c, 1, 1)
    shapes["model.cluster1.0.bias"] = (d_out,)
    shapes["model.cluster2.0.weight"] = (c, c, 1, 1)
    shapes["model.cluster2.0.bias"] = (c,)
    shapes["model.cluster2.2.weight"] = (d_out, c, 1, 1)
    shapes["model.cluster2.2.bias"] = (d_out,)
    if num_pq == 1:
        shapes["model.vq_blocks.0.embedding.weight"] = (k, d_out)
    else:
        for i in range(num_pq):
            shapes[f"model.vq_blocks.0.quantizers.{i}.embedding.weight"] = (k, d_out // num_pq)
    shapes["evaluator.linear_probe.weight"] = (n_cls, d_out, 1, 1)
    shapes["evaluator.linear_probe.bias"] = (n_cls,)
    shapes["evaluator.cluster_probe.clusters"] = (n_cls + extra, d_out)
    return shapes


def _draw(rng, key, shape, num_codebook):
    n = int(np.prod(shape))
    if key.endswith("embedding.weight"):
        lim = 1.0 / num_codebook  # model/dino_pqgo.py:493
        arr = rng.uniform(-lim, lim, size=n)
    elif "norm" in key and key.endswith(".weight") and len(shape) == 1:
        arr = 1.0 + 0.1 * rng.standard_normal(n)
    elif key.endswith(".bias"):
        arr = 0.05 * rng.standard_normal(n)
    elif key.endswith("cls_token") or key.endswith("pos_embed"):
        arr = 0.2 * rng.standard_normal(n)
    elif key.endswith("cluster_probe.clusters"):
        arr = rng.standard_normal(n)  # model/evaluator.py:91
    elif key.endswith("mlp.fc2.weight") or key.endswith("attn.proj.weight"):
        arr = 0.02 * rng.standard_normal(n)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        arr = rng.standard_normal(n) / math.sqrt(fan_in)
    return torch.from_numpy(arr.astype(np.float32).reshape(shape))


def synth_pqgo_state(cfg, seed=1234):
    """Full PQGOWrapper state_dict drawn from PCG64(seed) in sorted key order."""
    shapes = wrapper_state_shapes(cfg)
    k = cfg["model"]["vq"]["num_codebooks"][0]
    rng = np.random.Generator(np.random.PCG64(seed))
    state = OrderedDict()
    drawn = {key: _draw(rng, key, shapes[key], k) for key in sorted(shapes)}
    for key in shapes:  # keep module order for load_state_dict readability
        state[key] = drawn[key]
    return state


def synth_backbone_state(cfg, seed=1234):
    """Backbone-only view of synth_pqgo_state with the prefix stripped: what a DINO
    ``{"teacher": state_dict}`` checkpoint holds (model/dino/dino_featurizer.py:38-46)."""
    full = synth_pqgo_state(cfg, seed)
    pre = "model.extractor.backbone."
    return OrderedDict((k[len(pre):], v) for k, v in full.items() if k.startswith(pre))


def synth_images(batch, height, width, seed=0):
    """ImageNet-normalised-looking images: N(0,1) fp32 (batch,3,H,W)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal((batch, 3, height, width), dtype=np.float32))


def synth_labels(batch, height, width, num_classes, seed=0):
    """Labels in [-1, num_classes): -1 is the ignore value (model/evaluator.py:73)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    return torch.from_numpy(rng.integers(-1, num_classes, size=(batch, height, width), dtype=np.int64))
