"""DINO ViT backbone and featurizer with the reference's module tree and state_dict keys
(model/dino/vision_transformer.py, model/dino/dino_featurizer.py), computed by the HIP kernels.

The nn.Modules here only *hold* parameters under the reference's names (so reference and DINO
checkpoints load with strict=True); the forward pass is one call into equss_vit_forward, which
runs patch embedding, the 12 blocks and the final norm as hand-written gfx950 kernels in bf16 with
fp32 accumulation and an fp32 residual stream.
"""
import math
import sys
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native
from .configs import VIT_ARCH

__all__ = ["VisionTransformer", "DinoFeaturizer", "vit_tiny", "vit_small", "vit_base"]


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    """Parameter container with the key names of vision_transformer.py:136-167; inference runs in
    HIP.  Drop-path / dropout rates of the reference are 0 on this path and are not modelled."""

    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, norm_layer=nn.LayerNorm, **_unused):
        super().__init__()
        if embed_dim // num_heads != 64:
            raise ValueError("the HIP attention kernel is built for head_dim 64 (vit_small / vit_base)")
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads = depth, num_heads
        self.patch_embed = _PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self._handle = None
        self._handle_key = None
        self._handle32 = None
        self._handle32_key = None
        self._pos_cache = {}
        # "bf16": bf16 MFMA operands, fp32 accumulation (the throughput path); "fp32": every product and sum
        # in fp32 (parity mode: comparable with the reference's fp32 CPU run element by element)
        self.precision = "bf16"

    # -- position embedding -----------------------------------------------------------------------
    def interpolate_pos_encoding(self, npatch, w, h):
        """vision_transformer.py:178-198.  Input independent, so it is evaluated once per image size
        (on the host, with the same torch call as the reference) and cached on the device."""
        n = self.pos_embed.shape[1] - 1
        if npatch == n and w == h:
            return self.pos_embed
        pos = self.pos_embed.detach().float().cpu()
        dim = pos.shape[-1]
        p = self.patch_embed.patch_size
        w0, h0 = w // p + 0.1, h // p + 0.1
        side = int(math.sqrt(n))
        grid = F.interpolate(pos[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2),
                             scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)), mode="bicubic")
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((pos[:, :1], grid), dim=1).to(self.pos_embed.device)

    def _pos_for(self, w, h):
        p = self.patch_embed.patch_size
        key = (w, h, self.pos_embed._version, self.pos_embed.data_ptr())
        hit = self._pos_cache.get("pos")
        if hit is None or hit[0] != key:
            pos = self.interpolate_pos_encoding((w // p) * (h // p), w, h)
            hit = (key, pos.detach().reshape(-1, self.embed_dim).contiguous().float())
            self._pos_cache["pos"] = hit
        return hit[1]

    def invalidate_caches(self) -> None:
        """Forget the device weight set and the resized position embedding (rebuilt on the next forward).
        Only needed after parameters were written through ``.data``, which leaves their version counters
        alone; ordinary in-place writes, load_state_dict and .to() are noticed without it."""
        self._handle = None
        self._handle_key = None
        self._handle32 = None
        self._handle32_key = None
        self._pos_cache = {}

    # -- device weight set ------------------------------------------------------------------------
    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def device_handle(self, fp32=False):
        """bf16 (or, for the parity mode, fp32) GEMM weights + fp32 norm/bias vectors in the layout the
        C ABI wants; rebuilt only when a parameter was modified or moved (the backbone is frozen on this path)."""
        key = self._weights_key()
        if fp32:
            if self._handle32 is None or self._handle32_key != key:
                self._handle32 = self._build_handle(torch.float32)
                self._handle32_key = key
            return self._handle32
        if self._handle is None or self._handle_key != key:
            self._handle = self._build_handle(torch.bfloat16)
            self._handle_key = key
        return self._handle

    def _build_handle(self, gemm_dtype):
        dev = self.pos_embed.device
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()        # noqa: E731
        b16 = lambda t: t.detach().to(dev, gemm_dtype).contiguous()   # noqa: E731  (GEMM operands)
        c = self.embed_dim
        t = {"patch_w": b16(self.patch_embed.proj.weight.reshape(c, -1)),
             "patch_b": f32(self.patch_embed.proj.bias),
             "cls": f32(self.cls_token.reshape(-1)),
             "norm_g": f32(self.norm.weight), "norm_b": f32(self.norm.bias)}
        for i, blk in enumerate(self.blocks):
            pre = f"blocks.{i}."
            qkv_b = blk.attn.qkv.bias if blk.attn.qkv.bias is not None else torch.zeros(3 * c)
            t[pre + "ln1_g"], t[pre + "ln1_b"] = f32(blk.norm1.weight), f32(blk.norm1.bias)
            t[pre + "qkv_w"], t[pre + "qkv_b"] = b16(blk.attn.qkv.weight), f32(qkv_b)
            t[pre + "proj_w"], t[pre + "proj_b"] = b16(blk.attn.proj.weight), f32(blk.attn.proj.bias)
            t[pre + "ln2_g"], t[pre + "ln2_b"] = f32(blk.norm2.weight), f32(blk.norm2.bias)
            t[pre + "fc1_w"], t[pre + "fc1_b"] = b16(blk.mlp.fc1.weight), f32(blk.mlp.fc1.bias)
            t[pre + "fc2_w"], t[pre + "fc2_b"] = b16(blk.mlp.fc2.weight), f32(blk.mlp.fc2.bias)
        return native.VitHandle(t, c, self.depth, self.num_heads, self.patch_embed.patch_size)

    # -- forward ----------------------------------------------------------------------------------
    @torch.no_grad()
    def patch_features(self, img, feat_bf16=None, want_f32=False):
        """Final-norm patch tokens (CLS dropped), token-major: (B*n, C) bf16 [and fp32]."""
        img = img.detach().to(torch.float32).contiguous()
        _, _, h, w = img.shape
        pos = self._pos_for(h, w)  # the reference names dims 2,3 "w, h"; the maths is symmetric in them
        if self.precision == "fp32":
            f32 = native.vit_forward_f32(self.device_handle(fp32=True), img, pos)
            if feat_bf16 is not None:
                feat_bf16.copy_(f32)
            else:
                feat_bf16 = f32.to(torch.bfloat16)
            return (feat_bf16, f32) if want_f32 else feat_bf16
        return native.vit_forward(self.device_handle(), img, pos, feat_bf16=feat_bf16, want_f32=want_f32)

    @torch.no_grad()
    def get_intermediate_feat(self, x, n=1):
        """vision_transformer.py:227-239: ([norm(x) after each of the last n blocks, (B, 1+n_patches, C)
        fp32 with the CLS row], [], []).  The attention maps and qkv the reference also returns are never
        formed by the fused attention kernel (its only caller drops them, dino_featurizer.py:76-80), so
        those two lists stay empty."""
        if not 1 <= n <= self.depth:
            raise ValueError(f"n={n} must lie in [1, {self.depth}]")
        img = x.detach().to(torch.float32).contiguous()
        _, _, h, w = img.shape
        if self.precision == "fp32" and n == 1:
            b = img.shape[0]
            feat = native.vit_forward_f32(self.device_handle(fp32=True), img, self._pos_for(h, w), keep_cls=True)
            return [feat.reshape(b, -1, self.embed_dim)], [], []
        if self.precision == "fp32":
            # (no silent change of precision: the fp32 parity forward keeps only the last block's features)
            raise NotImplementedError("precision='fp32' returns the last block's features only (n=1); n > 1 runs in bf16")
        feats = native.vit_intermediate_feat(self.device_handle(), img, self._pos_for(h, w), n_last=n)
        return [feats[i] for i in range(n)], [], []

    @torch.no_grad()
    def residual_taps(self, x):
        """The fp32 residual stream after every block, (depth, B, 1+n_patches, C): for parity checks
        against per-block reference outputs.  Always from the bf16 forward (bf16 GEMM operands, fp32 residual stream),
        whatever ``precision`` says: the fp32 parity forward has no taps."""
        img = x.detach().to(torch.float32).contiguous()
        _, _, h, w = img.shape
        _feats, taps = native.vit_intermediate_feat(self.device_handle(), img, self._pos_for(h, w), n_last=1, want_taps=True)
        return taps

    def forward(self, x):
        raise NotImplementedError("the CLS-token classifier output is not part of the EQUSS path")


_LN = partial(nn.LayerNorm, eps=1e-6)


def vit_tiny(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4,
                             qkv_bias=True, norm_layer=_LN, **kw)


def vit_small(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4,
                             qkv_bias=True, norm_layer=_LN, **kw)


def vit_base(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                             qkv_bias=True, norm_layer=_LN, **kw)


_ARCH = {"vit_tiny": vit_tiny, "vit_small": vit_small, "vit_base": vit_base}


class DinoFeaturizer(nn.Module):
    """model/dino/dino_featurizer.py:8-83 with the same constructor config, attributes
    (``backbone``, ``n_feats``, ``patch_size``, ``freeze_backbone``) and forward contract."""

    def __init__(self, cfg: dict):
        super().__init__()
        self.cfg = cfg
        arch = cfg["model_type"]
        patch_size = cfg["dino_patch_size"]
        self.patch_size = patch_size
        self.freeze_backbone = cfg.get("freeze_backbone", True)
        if arch not in ("vit_small", "vit_base") or patch_size not in (8, 16):
            raise ValueError(f"Unknown arch {arch} and patch size {patch_size}.")
        self.backbone = _ARCH[arch](patch_size=patch_size, num_classes=0)
        self.backbone.requires_grad_(not self.freeze_backbone)
        self.backbone.eval()
        if not self.freeze_backbone:
            raise NotImplementedError("fine-tuning the backbone needs a ViT backward, which the HIP path does not have")
        weights = cfg.get("pretrained_weights")
        if weights is not None:
            state = torch.load(weights, map_location="cpu", weights_only=True)["teacher"]
            state = {k.replace("module.", "").replace("backbone.", ""): v for k, v in state.items()}
            msg = self.backbone.load_state_dict(state, strict=False)
            print(f"Pretrained weights found at {weights} and loaded with msg: {msg}")
        else:
            # The reference downloads the DINO checkpoint here (dino_featurizer.py:48-51).  Fetching
            # is the caller's business on an air-gapped node: load a state_dict afterwards.
            print("No pretrained_weights given: backbone left at its random initialisation; "
                  "load a DINO state_dict (same key names) before use.", file=sys.stderr)
        self.n_feats = VIT_ARCH[arch][0]

    def train(self, mode: bool = True):
        super().train(mode=mode)
        if self.freeze_backbone:
            self.backbone.eval()
        return self

    @torch.no_grad()
    def forward_tokens(self, img, feat_bf16=None, want_f32=False):
        b, c, h, w = img.shape
        assert (h % self.patch_size == 0) and (w % self.patch_size == 0)
        return self.backbone.patch_features(img, feat_bf16=feat_bf16, want_f32=want_f32)

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        """(b, 3, H, W) -> (b, C, H/p, W/p) fp32.  The values live token-major in memory; the
        returned tensor is that buffer viewed channels-first (same shape and values as the
        reference's contiguous NCHW result)."""
        b, c, h, w = img.shape
        _, f32 = self.forward_tokens(img, want_f32=True)
        return f32.reshape(b, h // self.patch_size, w // self.patch_size, -1).permute(0, 3, 1, 2)
