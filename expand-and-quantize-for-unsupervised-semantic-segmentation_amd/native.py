"""ctypes binding of libequss_hip.so (the C ABI declared in include/equss_hip.h) plus thin
tensor-level wrappers.  PyTorch only supplies device memory and the current stream here; every
computation below runs in the hand-written gfx950 kernels.  There is no fallback: if the library
is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import threading

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libequss_hip.so")

NORM_MODES = {"none": 0, "l2": 1, "z_norm": 2}
EPI_BF16, EPI_GELU_BF16, EPI_RELU_BF16, EPI_F32, EPI_RESID_F32 = 0, 1, 2, 3, 4

DEV_SYMBOLS = [   # include/equss_hip_dev.h: developer knobs, not part of the drop-in ABI
    "equss_dev_pq_mode", "equss_dev_gemm_variant", "equss_dev_gemm_stages", "equss_dev_ws_bk", "equss_dev_gemm_cu_share",
    "equss_dev_attn_diag", "equss_dev_mlp_fused", "equss_dev_mlp_tail_max", "equss_dev_fuse_min_tiles", "equss_dev_tail_images", "equss_dev_block_ws",
    "equss_dev_rs_stagger",
]

EXPORTED_SYMBOLS = [
    "equss_abi_version", "equss_last_error", "equss_profile_enable", "equss_profile_read",
    "equss_pq_workspace_bytes", "equss_pq_assign", "equss_pq_prob",
    "equss_pq_static_scratch_bytes", "equss_pq_assign_static",
    "equss_layernorm_bf16", "equss_gemm_bf16", "equss_gemm_resid_ln_bf16", "equss_mlp_resid_ln_bf16", "equss_block_tail_bf16", "equss_attention_bf16",
    "equss_patch_workspace_bytes", "equss_patch_tokens",
    "equss_vit_workspace_bytes", "equss_vit_forward", "equss_vit_intermediate_feat", "equss_expand_head",
    "equss_vit_tail_image_bytes", "equss_vit_pack_tail_images", "equss_code_sums",
    "equss_vit_f32_workspace_bytes", "equss_vit_forward_f32", "equss_expand_head_f32",
    "equss_probe_workspace_bytes", "equss_probe_eval", "equss_probe_train", "equss_confusion_update", "equss_pq_st_backward", "equss_gemm_tn_bf16", "equss_tap_gather", "equss_tap_scatter",
    "equss_index_rows_i16", "equss_code_usage",
]


ABI_VERSION = 2   # EQUSS_ABI_VERSION of include/equss_hip.h this binding is written against


class EqussError(RuntimeError):
    pass


class VitBlockStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
        "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VitStruct(ctypes.Structure):
    _fields_ = [("embed_dim", ctypes.c_int32), ("depth", ctypes.c_int32), ("heads", ctypes.c_int32),
                ("patch", ctypes.c_int32),
                ("patch_w", ctypes.c_void_p), ("patch_b", ctypes.c_void_p), ("cls", ctypes.c_void_p),
                ("norm_g", ctypes.c_void_p), ("norm_b", ctypes.c_void_p),
                ("blocks", ctypes.POINTER(VitBlockStruct)), ("tail_images", ctypes.c_void_p)]


_lib = None
_lock = threading.Lock()


def _declare(lib):
    c_int, c_i64, c_vp, c_sz, c_f = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float
    lib.equss_abi_version.restype = c_int
    lib.equss_abi_version.argtypes = []
    lib.equss_last_error.restype = ctypes.c_char_p
    lib.equss_last_error.argtypes = []
    lib.equss_profile_enable.restype = c_int
    lib.equss_profile_enable.argtypes = [c_int]
    lib.equss_profile_read.restype = c_int
    lib.equss_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64), c_int]
    lib.equss_pq_workspace_bytes.restype = c_sz
    lib.equss_pq_workspace_bytes.argtypes = [c_int] * 4
    lib.equss_pq_assign.restype = c_int
    lib.equss_pq_assign.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_pq_static_scratch_bytes.restype = c_sz
    lib.equss_pq_static_scratch_bytes.argtypes = []
    lib.equss_pq_assign_static.restype = c_int
    lib.equss_pq_assign_static.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp,
                                           c_int, c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_pq_prob.restype = c_int
    lib.equss_pq_prob.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_int, c_int, c_int, c_int, c_int, c_f, c_vp]
    lib.equss_layernorm_bf16.restype = c_int
    lib.equss_layernorm_bf16.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_f, c_vp]
    lib.equss_gemm_bf16.restype = c_int
    lib.equss_gemm_bf16.argtypes = [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_gemm_resid_ln_bf16.restype = c_int
    lib.equss_gemm_resid_ln_bf16.argtypes = [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64,
                                             c_int, c_int, c_int, c_f, c_vp]
    lib.equss_mlp_resid_ln_bf16.restype = c_int
    lib.equss_mlp_resid_ln_bf16.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64,
                                            c_int, c_int, c_int, c_f, c_vp]
    lib.equss_block_tail_bf16.restype = c_int
    lib.equss_block_tail_bf16.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                          c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_f, c_vp]
    lib.equss_attention_bf16.restype = c_int
    lib.equss_attention_bf16.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_f, c_vp]
    lib.equss_patch_workspace_bytes.restype = c_sz
    lib.equss_patch_workspace_bytes.argtypes = [c_int] * 5
    lib.equss_patch_tokens.restype = c_int
    lib.equss_patch_tokens.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                                       c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_vit_workspace_bytes.restype = c_sz
    lib.equss_vit_workspace_bytes.argtypes = [ctypes.POINTER(VitStruct), c_int, c_int, c_int]
    lib.equss_vit_tail_image_bytes.restype = c_sz
    lib.equss_vit_tail_image_bytes.argtypes = [ctypes.POINTER(VitStruct)]
    lib.equss_vit_pack_tail_images.restype = c_int
    lib.equss_vit_pack_tail_images.argtypes = [ctypes.POINTER(VitStruct), c_vp, c_sz, c_vp]
    lib.equss_vit_forward.restype = c_int
    lib.equss_vit_forward.argtypes = [ctypes.POINTER(VitStruct), c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_sz,
                                      c_int, c_int, c_int, c_vp]
    lib.equss_vit_intermediate_feat.restype = c_int
    lib.equss_vit_intermediate_feat.argtypes = [ctypes.POINTER(VitStruct), c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_sz,
                                                c_int, c_int, c_int, c_vp]
    lib.equss_vit_f32_workspace_bytes.restype = c_sz
    lib.equss_vit_f32_workspace_bytes.argtypes = [ctypes.POINTER(VitStruct), c_int, c_int, c_int]
    lib.equss_vit_forward_f32.restype = c_int
    lib.equss_vit_forward_f32.argtypes = [ctypes.POINTER(VitStruct), c_vp, c_vp, c_vp, c_int, c_vp, c_sz,
                                          c_int, c_int, c_int, c_vp]
    lib.equss_expand_head_f32.restype = c_int
    lib.equss_expand_head_f32.argtypes = [c_vp] * 9 + [c_int, c_int, c_int, c_vp]
    lib.equss_expand_head.restype = c_int
    lib.equss_expand_head.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]
    lib.equss_probe_workspace_bytes.restype = c_sz
    lib.equss_probe_workspace_bytes.argtypes = [c_int] * 5
    lib.equss_confusion_update.restype = c_int
    lib.equss_confusion_update.argtypes = [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]
    lib.equss_index_rows_i16.restype = c_int
    lib.equss_index_rows_i16.argtypes = [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_code_usage.restype = c_int
    lib.equss_code_usage.argtypes = [c_vp, c_vp, c_i64, c_int, c_int, c_vp]
    lib.equss_code_sums.restype = c_int
    lib.equss_code_sums.argtypes = [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]
    lib.equss_tap_gather.restype = c_int
    lib.equss_tap_gather.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_i64, c_vp]
    lib.equss_tap_scatter.restype = c_int
    lib.equss_tap_scatter.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_i64, c_vp]
    lib.equss_gemm_tn_bf16.restype = c_int
    lib.equss_gemm_tn_bf16.argtypes = [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp]
    lib.equss_pq_st_backward.restype = c_int
    lib.equss_pq_st_backward.argtypes = [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, ctypes.c_float, ctypes.c_float,
                                         c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_probe_eval.restype = c_int
    lib.equss_probe_eval.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]
    lib.equss_probe_train.restype = c_int
    lib.equss_probe_train.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]


def load_library():
    """dlopen libequss_hip.so (once) and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise EqussError(
                    f"{LIB_PATH} is not built. Run `python __graft_entry__.py build` (hipcc, gfx950). "
                    "This package has no CPU or eager fallback.")
            lib = ctypes.CDLL(LIB_PATH)
            for sym in EXPORTED_SYMBOLS:
                if not hasattr(lib, sym):
                    raise EqussError(f"{LIB_PATH} does not export {sym}")
            _declare(lib)
            if lib.equss_abi_version() != ABI_VERSION:
                raise EqussError("libequss_hip.so ABI version mismatch")
            _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().equss_last_error().decode("utf-8", "replace")
        if rc in (2, 3, 4):
            raise ValueError(f"{what}: {msg} (code {rc})")
        raise EqussError(f"{what}: {msg} (code {rc})")


def _stream(t):
    """The current HIP stream of ``t``'s device.  Kernels launch on the CURRENT device, so that device
    must be the tensor's: a launch with a foreign device's pointers and stream handle would corrupt
    memory silently, hence the loud failure (one process per GPU, as torchrun sets it up, never trips it)."""
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise EqussError(f"tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                         f"wrap the call in `with torch.cuda.device({t.device.index}):`")
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise EqussError("the EQUSS HIP path needs tensors on a ROCm device (no CPU fallback exists)")


_ws_cache = {}


def _workspace(device, nbytes, tag):
    """A per-(device, stream, tag) scratch tensor that only grows; 256-byte aligned by the caching
    allocator.  Keyed by the current stream as well: launches on different streams may overlap."""
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


KERNEL_CLASSES = ["patch", "layernorm", "gemm_qkv", "attention", "gemm_proj", "gemm_fc1", "gemm_fc2",
                  "head_gemm1", "head_gemm2", "pq_prep", "pq_assign", "probe_table", "probe_pixels", "mlp_fused", "block_tail",
                  "pack_weights"]


def profile_enable(max_pairs):
    """Start (max_pairs > 0) or stop (0) recording a HIP event pair around every kernel-class launch."""
    _check(load_library().equss_profile_enable(int(max_pairs)), "equss_profile_enable")


def profile_read(reset=True):
    """{class: (total_ms, launches)} of everything recorded since the last reset (synchronises)."""
    n = len(KERNEL_CLASSES)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int64 * n)()
    _check(load_library().equss_profile_read(ms, cnt, int(bool(reset))), "equss_profile_read")
    return {k: (ms[i], cnt[i]) for i, k in enumerate(KERNEL_CLASSES)}


# ------------------------------------------------------------------------------------------------
# Product quantisation
# ------------------------------------------------------------------------------------------------
def pq_assign(code_rows, codebooks, normalize="l2", gather_normalized=False, want_zq=True, want_loss=True, zq_out=None):
    """code_rows (rows, num_pq*d) fp32 (row stride may exceed the width), codebooks (num_pq, K, d) fp32.
    Returns idx (num_pq, rows) int64, zq (rows, num_pq*d) fp32 or None, loss_sum (num_pq,) fp32 or None.
    zq_out: optional contiguous (rows, num_pq*d) fp32 destination for zq (e.g. a row range of a larger buffer)."""
    lib = load_library()
    _need_cuda(code_rows, codebooks)
    if code_rows.dtype != torch.float32 or codebooks.dtype != torch.float32:
        raise ValueError("pq_assign computes in fp32: code and codebooks must be float32")
    if normalize not in NORM_MODES:
        raise ValueError(f"Unsupported normalize type {normalize}")
    num_pq, k, d = codebooks.shape
    rows, width = code_rows.shape
    if width != num_pq * d:
        raise ValueError(f"code has {width} channels, codebooks need {num_pq}x{d}")
    if code_rows.stride(1) != 1 or code_rows.stride(0) % 4 != 0 or code_rows.data_ptr() % 16 != 0:
        code_rows = code_rows.contiguous()
    codebooks = codebooks.contiguous()
    dev = code_rows.device
    idx = torch.empty((num_pq, rows), dtype=torch.int64, device=dev)
    if zq_out is not None:
        if zq_out.shape != (rows, width) or zq_out.dtype != torch.float32 or not zq_out.is_contiguous():
            raise ValueError("zq_out must be a contiguous float32 (rows, num_pq*d) tensor")
        zq = zq_out
    else:
        zq = torch.empty((rows, width), dtype=torch.float32, device=dev) if want_zq else None
    loss = torch.empty((num_pq,), dtype=torch.float32, device=dev) if want_loss else None
    if rows == 0:
        if loss is not None:
            loss.zero_()
        return idx, zq, loss
    nbytes = lib.equss_pq_workspace_bytes(rows, num_pq, k, d)
    ws = _workspace(dev, nbytes, "pq")
    rc = lib.equss_pq_assign(_ptr(code_rows), code_rows.stride(0), _ptr(codebooks), _ptr(idx), _ptr(zq),
                             width, _ptr(loss), _ptr(ws), ws.numel(), rows, num_pq, k, d,
                             NORM_MODES[normalize], int(bool(gather_normalized)), _stream(code_rows))
    _check(rc, "equss_pq_assign")
    return idx, zq, loss


def pq_prob(code_rows, codebooks, normalize="l2", tau=1.0):
    """softmax(-dist / tau) over the K codewords: (rows, num_pq*K) fp32."""
    lib = load_library()
    _need_cuda(code_rows, codebooks)
    if code_rows.dtype != torch.float32 or codebooks.dtype != torch.float32:
        raise ValueError("pq_prob computes in fp32: code and codebooks must be float32")
    num_pq, k, d = codebooks.shape
    rows, width = code_rows.shape
    if width != num_pq * d:
        raise ValueError(f"code has {width} channels, codebooks need {num_pq}x{d}")
    if code_rows.stride(1) != 1 or code_rows.stride(0) % 4 != 0 or code_rows.data_ptr() % 16 != 0:
        code_rows = code_rows.contiguous()
    codebooks = codebooks.contiguous()
    prob = torch.empty((rows, num_pq * k), dtype=torch.float32, device=code_rows.device)
    if rows == 0:
        return prob
    nbytes = lib.equss_pq_workspace_bytes(rows, num_pq, k, d)
    ws = _workspace(code_rows.device, nbytes, "pq")
    rc = lib.equss_pq_prob(_ptr(code_rows), code_rows.stride(0), _ptr(codebooks), _ptr(prob), _ptr(ws), ws.numel(),
                           rows, num_pq, k, d, NORM_MODES[normalize], 1.0 / float(tau), _stream(code_rows))
    _check(rc, "equss_pq_prob")
    return prob


# ------------------------------------------------------------------------------------------------
# ViT building blocks
# ------------------------------------------------------------------------------------------------
def layernorm_bf16(x, gamma, beta, eps=1e-6, tokens_per_image=0, out=None, want_f32=False):
    """x (rows, C) fp32 -> bf16 (rows_out, C); with tokens_per_image > 0 the CLS row of every image is dropped."""
    lib = load_library()
    _need_cuda(x, gamma, beta)
    rows, c = x.shape
    rows_out = rows if tokens_per_image <= 0 else rows - rows // tokens_per_image
    if out is None:
        out = torch.empty((rows_out, c), dtype=torch.bfloat16, device=x.device)
    out_f32 = torch.empty((rows_out, c), dtype=torch.float32, device=x.device) if want_f32 else None
    rc = lib.equss_layernorm_bf16(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(out), out.stride(0),
                                  _ptr(out_f32), rows, c, int(tokens_per_image), float(eps), _stream(x))
    _check(rc, "equss_layernorm_bf16")
    return (out, out_f32) if want_f32 else out


def gemm_bf16(a, w, bias, epilogue, out=None):
    """out = epilogue(a @ w.T + bias); a (M,K) bf16, w (N,K) bf16, bias (N,) fp32 or None."""
    lib = load_library()
    _need_cuda(a, w, bias)
    m, k = a.shape
    n = w.shape[0]
    if out is None:
        dt = torch.float32 if epilogue in (EPI_F32, EPI_RESID_F32) else torch.bfloat16
        if epilogue == EPI_RESID_F32:
            raise ValueError("the residual epilogue accumulates into `out`; pass it")
        out = torch.empty((m, n), dtype=dt, device=a.device)
    rc = lib.equss_gemm_bf16(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(out), out.stride(0),
                             m, n, k, int(epilogue), _stream(a))
    _check(rc, "equss_gemm_bf16")
    return out


def gemm_resid_ln_bf16(a, w, bias, x, gamma=None, beta=None, eps=1e-6, want_ln=True):
    """x += a @ w.T + bias (fp32, in place); returns LayerNorm(x) * gamma + beta in bf16 (or None)."""
    lib = load_library()
    _need_cuda(a, w, bias, x, gamma, beta)
    m, k = a.shape
    n = w.shape[0]
    ln = torch.empty((m, n), dtype=torch.bfloat16, device=a.device) if want_ln else None
    rc = lib.equss_gemm_resid_ln_bf16(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(x), x.stride(0),
                                      _ptr(gamma), _ptr(beta), _ptr(ln), n if ln is None else ln.stride(0), m, n, k,
                                      float(eps), _stream(a))
    _check(rc, "equss_gemm_resid_ln_bf16")
    return ln


def mlp_resid_ln_bf16(a, w1, b1, w2, b2, x, gamma=None, beta=None, eps=1e-6, want_ln=True):
    """x += gelu(a @ w1.T + b1) @ w2.T + b2 (fp32, in place); returns LayerNorm(x) * gamma + beta in bf16 (or None)."""
    lib = load_library()
    _need_cuda(a, w1, b1, w2, b2, x, gamma, beta)
    m, c = a.shape
    hidden = w1.shape[0]
    ln = torch.empty((m, c), dtype=torch.bfloat16, device=a.device) if want_ln else None
    rc = lib.equss_mlp_resid_ln_bf16(_ptr(a), a.stride(0), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(x), x.stride(0),
                                     _ptr(gamma), _ptr(beta), _ptr(ln), c if ln is None else ln.stride(0), m, c, hidden,
                                     float(eps), _stream(a))
    _check(rc, "equss_mlp_resid_ln_bf16")
    return ln


def block_tail_bf16(att, wp, bp, gamma2, beta2, w1, b1, w2, b2, x, gamma=None, beta=None, eps=1e-6, want_ln=True):
    """x += att @ wp.T + bp; x += gelu(LN2(x) @ w1.T + b1) @ w2.T + b2 (fp32, in place); returns
    (LayerNorm2 of the intermediate x, LayerNorm(x) * gamma + beta or None), both bf16."""
    lib = load_library()
    _need_cuda(att, wp, bp, gamma2, beta2, w1, b1, w2, b2, x, gamma, beta)
    m, c = att.shape
    hidden = w1.shape[0]
    mid = torch.empty((m, c), dtype=torch.bfloat16, device=att.device)
    ln = torch.empty((m, c), dtype=torch.bfloat16, device=att.device) if want_ln else None
    rc = lib.equss_block_tail_bf16(_ptr(att), att.stride(0), _ptr(wp), _ptr(bp), _ptr(gamma2), _ptr(beta2), _ptr(mid),
                                   mid.stride(0), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(x), x.stride(0), _ptr(gamma),
                                   _ptr(beta), _ptr(ln), c if ln is None else ln.stride(0), m, c, hidden, float(eps),
                                   _stream(att))
    _check(rc, "equss_block_tail_bf16")
    return mid, ln


def attention_bf16(qkv, batch, tokens, heads, scale, out=None):
    """qkv (batch*tokens, 3*heads*64) bf16 -> (batch*tokens, heads*64) bf16."""
    lib = load_library()
    _need_cuda(qkv)
    if out is None:
        out = torch.empty((batch * tokens, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    rc = lib.equss_attention_bf16(_ptr(qkv), _ptr(out), batch, tokens, heads, float(scale), _stream(qkv))
    _check(rc, "equss_attention_bf16")
    return out


def patch_tokens(img, w_patch, b_patch, cls, pos, patch):
    """img (B,3,H,W) fp32 -> x (B*(1+n), C) fp32 = cat(cls, patch_embed(img)) + pos."""
    lib = load_library()
    _need_cuda(img, w_patch, b_patch, cls, pos)
    b, _, h, w = img.shape
    c = w_patch.shape[0]
    n = (h // patch) * (w // patch)
    x = torch.empty((b * (n + 1), c), dtype=torch.float32, device=img.device)
    nbytes = lib.equss_patch_workspace_bytes(b, h, w, patch, c)
    ws = _workspace(img.device, nbytes, "patch")
    rc = lib.equss_patch_tokens(_ptr(img), _ptr(w_patch), _ptr(b_patch), _ptr(cls), _ptr(pos), _ptr(x), _ptr(ws),
                                ws.numel(), b, h, w, patch, c, _stream(img))
    _check(rc, "equss_patch_tokens")
    return x


class VitHandle:
    """Device-resident bf16/fp32 weight set of one backbone plus the host-side EqussVit struct."""

    def __init__(self, tensors, embed_dim, depth, heads, patch, pack_tail=True):
        self.tensors = tensors  # keeps the device buffers alive
        self.blocks = (VitBlockStruct * depth)()
        for i in range(depth):
            for name, _ in VitBlockStruct._fields_:
                setattr(self.blocks[i], name, tensors[f"blocks.{i}.{name}"].data_ptr())
        self.struct = VitStruct(embed_dim, depth, heads, patch,
                                tensors["patch_w"].data_ptr(), tensors["patch_b"].data_ptr(),
                                tensors["cls"].data_ptr(), tensors["norm_g"].data_ptr(),
                                tensors["norm_b"].data_ptr(), self.blocks, None)
        self.embed_dim, self.depth, self.heads, self.patch = embed_dim, depth, heads, patch
        # The block-tail kernel's weight order, written once per weight set (a handle is rebuilt when a parameter changes)
        # instead of at the start of every forward.  bf16 sets only: the fp32 parity forward has no such kernel.
        self.tail_images = None
        if pack_tail and tensors["patch_w"].is_cuda and tensors["blocks.0.proj_w"].dtype == torch.bfloat16:
            lib = load_library()
            nbytes = lib.equss_vit_tail_image_bytes(ctypes.byref(self.struct))
            if nbytes:
                self.tail_images = torch.empty(nbytes, dtype=torch.uint8, device=tensors["patch_w"].device)
                _check(lib.equss_vit_pack_tail_images(ctypes.byref(self.struct), _ptr(self.tail_images), nbytes,
                                                      _stream(self.tail_images)), "equss_vit_pack_tail_images")
                self.struct.tail_images = self.tail_images.data_ptr()


def vit_forward(handle, img, pos, feat_bf16=None, ld_feat=None, want_f32=False):
    """Patch features of ``img`` (B,3,H,W) fp32: returns feat (B*n, C) bf16 view (and fp32 if asked)."""
    lib = load_library()
    _need_cuda(img, pos)
    b, _, h, w = img.shape
    c = handle.embed_dim
    n = (h // handle.patch) * (w // handle.patch)
    if feat_bf16 is None:
        feat_bf16 = torch.empty((b * n, c), dtype=torch.bfloat16, device=img.device)
    ld = feat_bf16.stride(0) if ld_feat is None else ld_feat
    feat_f32 = torch.empty((b * n, c), dtype=torch.float32, device=img.device) if want_f32 else None
    nbytes = lib.equss_vit_workspace_bytes(ctypes.byref(handle.struct), b, h, w)
    ws = _workspace(img.device, nbytes, "vit")
    rc = lib.equss_vit_forward(ctypes.byref(handle.struct), _ptr(img), _ptr(pos), _ptr(feat_bf16), ld, _ptr(feat_f32),
                               _ptr(ws), ws.numel(), b, h, w, _stream(img))
    _check(rc, "equss_vit_forward")
    return (feat_bf16, feat_f32) if want_f32 else feat_bf16


def vit_intermediate_feat(handle, img, pos, n_last=1, want_taps=False):
    """norm(x) after each of the last ``n_last`` blocks, CLS row kept: (n_last, B, 1+n, C) fp32
    (vision_transformer.py:227-239); with want_taps also the residual stream after every block,
    (depth, B, 1+n, C) fp32."""
    lib = load_library()
    _need_cuda(img, pos)
    b, _, h, w = img.shape
    c = handle.embed_dim
    tokens = (h // handle.patch) * (w // handle.patch) + 1
    feats = torch.empty((n_last, b, tokens, c), dtype=torch.float32, device=img.device)
    taps = torch.empty((handle.depth, b, tokens, c), dtype=torch.float32, device=img.device) if want_taps else None
    nbytes = lib.equss_vit_workspace_bytes(ctypes.byref(handle.struct), b, h, w)
    ws = _workspace(img.device, nbytes, "vit")
    rc = lib.equss_vit_intermediate_feat(ctypes.byref(handle.struct), _ptr(img), _ptr(pos), _ptr(feats), int(n_last),
                                         _ptr(taps), _ptr(ws), ws.numel(), b, h, w, _stream(img))
    _check(rc, "equss_vit_intermediate_feat")
    return (feats, taps) if want_taps else feats


def vit_forward_f32(handle, img, pos, keep_cls=False):
    """fp32 parity mode: patch features (B*n, C) fp32, or all tokens (B*(1+n), C) with keep_cls.
    ``handle`` must hold fp32 GEMM weights (VisionTransformer.device_handle(fp32=True))."""
    lib = load_library()
    _need_cuda(img, pos)
    b, _, h, w = img.shape
    c = handle.embed_dim
    n = (h // handle.patch) * (w // handle.patch)
    feat = torch.empty((b * (n + 1 if keep_cls else n), c), dtype=torch.float32, device=img.device)
    nbytes = lib.equss_vit_f32_workspace_bytes(ctypes.byref(handle.struct), b, h, w)
    ws = _workspace(img.device, nbytes, "vit_f32")
    rc = lib.equss_vit_forward_f32(ctypes.byref(handle.struct), _ptr(img), _ptr(pos), _ptr(feat), int(bool(keep_cls)),
                                   _ptr(ws), ws.numel(), b, h, w, _stream(img))
    _check(rc, "equss_vit_forward_f32")
    return feat


def expand_head_f32(feat, w1, b1, w2a, b2a, w2b, b2b):
    """fp32 parity mode of the expansion head: feat (rows, C) fp32 -> code (rows, D) fp32."""
    lib = load_library()
    _need_cuda(feat, w1, b1, w2a, b2a, w2b, b2b)
    rows, c = feat.shape
    d = w1.shape[0]
    hidden = torch.empty((rows, c), dtype=torch.float32, device=feat.device)
    code = torch.empty((rows, d), dtype=torch.float32, device=feat.device)
    rc = lib.equss_expand_head_f32(_ptr(feat), _ptr(w1), _ptr(b1), _ptr(w2a), _ptr(b2a), _ptr(w2b), _ptr(b2b),
                                   _ptr(hidden), _ptr(code), rows, c, d, _stream(feat))
    _check(rc, "equss_expand_head_f32")
    return code


def expand_head(fh, w2a, b2a, w_cat, b_cat, out_dim, out=None):
    """fh (rows, 2C) bf16 with the features in columns [0,C): returns code (rows, D) fp32 (written into
    ``out`` when given: a contiguous (rows, D) fp32 tensor, e.g. a row range of a larger buffer)."""
    lib = load_library()
    _need_cuda(fh, w2a, b2a, w_cat, b_cat)
    rows, c2 = fh.shape
    if out is not None:
        if out.shape != (rows, out_dim) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 (rows, D) tensor")
        code = out
    else:
        code = torch.empty((rows, out_dim), dtype=torch.float32, device=fh.device)
    rc = lib.equss_expand_head(_ptr(fh), _ptr(w2a), _ptr(b2a), _ptr(w_cat), _ptr(b_cat), _ptr(code), rows, c2 // 2,
                               out_dim, _stream(fh))
    _check(rc, "equss_expand_head")
    return code


# ------------------------------------------------------------------------------------------------
# Probes of the eval loop
# ------------------------------------------------------------------------------------------------
PROBE_SLOTS = 32  # classes per probe the kernel's table holds


def pack_probes(linear_weight, linear_bias, clusters):
    """(64, D) fp32 probe matrix and (64,) bias for probe_eval: rows [0, C) are the linear probe's
    1x1-conv weights, rows [32, 32 + C') the clusters normalised as evaluator.py:95 does."""
    n_lin, n_clu = linear_weight.shape[0], clusters.shape[0]
    if n_lin > PROBE_SLOTS or n_clu > PROBE_SLOTS:
        raise ValueError(f"probe_eval holds at most {PROBE_SLOTS} classes per probe, got {n_lin} / {n_clu}")
    d = clusters.shape[1]
    probes = torch.zeros(2 * PROBE_SLOTS, d, dtype=torch.float32, device=clusters.device)
    probes[:n_lin] = linear_weight.reshape(n_lin, d).float()
    probes[PROBE_SLOTS:PROBE_SLOTS + n_clu] = torch.nn.functional.normalize(clusters.float(), dim=1)
    bias = torch.zeros(2 * PROBE_SLOTS, dtype=torch.float32, device=clusters.device)
    bias[:n_lin] = linear_bias.float()
    return probes, bias


def probe_eval(feat_rows, probes, bias, label, batch, h, w, n_linear, n_cluster, want_norm=False):
    """feat_rows (batch*h*w, D) fp32 token-major, label (batch, H, W) int64 ->
    (linear_preds, cluster_preds) (batch, H, W) int64 and losses (2,) fp32 = [linear CE, cluster loss];
    with want_norm also the (batch, H, W) fp32 norm of the up-sampled feature per pixel."""
    _need_cuda(feat_rows, probes, bias, label)
    if feat_rows.dtype != torch.float32 or feat_rows.dim() != 2 or feat_rows.shape[0] != batch * h * w:
        raise ValueError("probe_eval: feat_rows must be (batch*h*w, D) float32")
    if feat_rows.stride(1) != 1 or feat_rows.stride(0) % 4 or feat_rows.data_ptr() % 16:
        feat_rows = feat_rows.contiguous()
    if label.dtype != torch.int64 or label.dim() != 3 or label.shape[0] != batch:
        raise ValueError("probe_eval: label must be (batch, H, W) int64")
    label = label.contiguous()
    H, W = int(label.shape[1]), int(label.shape[2])
    dev = feat_rows.device
    lin = torch.empty((batch, H, W), dtype=torch.int64, device=dev)
    clu = torch.empty((batch, H, W), dtype=torch.int64, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    norm = torch.empty((batch, H, W), dtype=torch.float32, device=dev) if want_norm else None
    lib = load_library()
    nbytes = lib.equss_probe_workspace_bytes(batch, h, w, H, W)
    ws = _workspace(dev, nbytes, "probe")
    _check(lib.equss_probe_eval(_ptr(feat_rows), feat_rows.stride(0), _ptr(probes), _ptr(bias), _ptr(label),
                                _ptr(lin), _ptr(clu), _ptr(losses), _ptr(norm), _ptr(ws), ws.numel(),
                                batch, h, w, feat_rows.shape[1], H, W, n_linear, n_cluster, _stream(feat_rows)),
           "equss_probe_eval")
    if want_norm:
        return lin, clu, losses, norm
    return lin, clu, losses


def tap_gather(rows, idx, wgt):
    """out[p] = sum_k wgt[p, k] * rows[idx[p, k]]: rows (n, C) fp32, idx (P, 4) int64, wgt (P, 4) fp32 -> (P, C) fp32."""
    _need_cuda(rows, idx, wgt)
    if rows.stride(1) != 1 or rows.stride(0) % 4 or rows.data_ptr() % 16:
        rows = rows.contiguous()
    idx, wgt = idx.contiguous(), wgt.float().contiguous()
    out = torch.empty((idx.shape[0], rows.shape[1]), dtype=torch.float32, device=rows.device)
    _check(load_library().equss_tap_gather(_ptr(rows), rows.stride(0), _ptr(idx), _ptr(wgt), _ptr(out), out.stride(0),
                                           idx.shape[0], rows.shape[1], rows.shape[0], _stream(rows)), "equss_tap_gather")
    return out


def tap_scatter(g, idx, wgt, n_rows):
    """The backward of tap_gather: (n_rows, C) fp32 with grad[idx[p, k]] += wgt[p, k] * g[p]."""
    _need_cuda(g, idx, wgt)
    g = g.float()
    if g.stride(1) != 1:
        g = g.contiguous()
    idx, wgt = idx.contiguous(), wgt.float().contiguous()
    grad = torch.empty((n_rows, g.shape[1]), dtype=torch.float32, device=g.device)
    _check(load_library().equss_tap_scatter(_ptr(g), g.stride(0), _ptr(idx), _ptr(wgt), _ptr(grad), grad.stride(0),
                                            idx.shape[0], g.shape[1], n_rows, _stream(g)), "equss_tap_scatter")
    return grad


def gemm_tn_bf16(a, b):
    """a (rows, M) bf16, b (rows, N) bf16 (row-major, unit column stride) -> a^T b (M, N) fp32: the weight-gradient
    product of the training pass, summed over the rows on the matrix cores."""
    _need_cuda(a, b)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError("gemm_tn_bf16: a (rows, M) and b (rows, N) must be bfloat16 with equal row counts")

    def usable(t):
        return t if (t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0) else t.contiguous()
    a, b = usable(a), usable(b)
    out = torch.empty((a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
    _check(load_library().equss_gemm_tn_bf16(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0),
                                             a.shape[0], a.shape[1], b.shape[1], _stream(a)), "equss_gemm_tn_bf16")
    return out


def pq_st_backward(rows, emb, idx, g_zq, g_vq, beta, book, num_codes, d, want_rows=True, want_books=True):
    """Gradients of the training-pass quantisation (see equss_pq_st_backward in the header): rows / emb / g_zq
    (n, num_pq*d) fp32, idx (num_pq, n) int64, g_vq a 0-dim fp32 tensor -> (g_rows or None, g_books (num_pq, K, d) or None)."""
    _need_cuda(rows, emb, idx, g_vq)
    n, cols = rows.shape
    num_pq = cols // d

    def row_major(t):
        if t is None:
            return None
        t = t.float()
        if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            t = t.contiguous()
        return t
    rows, emb, g_zq = row_major(rows), row_major(emb), row_major(g_zq)
    idx = idx.contiguous()
    g_vq = g_vq.reshape(1).float().contiguous()
    g_rows = torch.empty((n, cols), dtype=torch.float32, device=rows.device) if want_rows else None
    g_books = torch.empty((num_pq, num_codes, d), dtype=torch.float32, device=rows.device) if want_books else None
    lib = load_library()
    _check(lib.equss_pq_st_backward(_ptr(rows), rows.stride(0), _ptr(emb), emb.stride(0), _ptr(idx), _ptr(g_zq),
                                    g_zq.stride(0) if g_zq is not None else 0, _ptr(g_vq), float(beta), float(book),
                                    _ptr(g_rows), g_rows.stride(0) if want_rows else 0, _ptr(g_books), n, num_pq,
                                    num_codes, d, _stream(rows)), "equss_pq_st_backward")
    return g_rows, g_books


def probe_train(feat_rows, probes, bias, label, batch, h, w, n_linear, n_cluster):
    """probe_eval for the training pass: (linear_preds, cluster_preds, losses (3,), grad_table (batch*h*w, 64)):
    losses = [linear CE, cluster loss, labelled pixels]; grad_table = gradients of the two loss sums with respect to
    the low-resolution logits / inner products (see equss_probe_train in the header)."""
    _need_cuda(feat_rows, probes, bias, label)
    if feat_rows.dtype != torch.float32 or feat_rows.dim() != 2 or feat_rows.shape[0] != batch * h * w:
        raise ValueError("probe_train: feat_rows must be (batch*h*w, D) float32")
    if feat_rows.stride(1) != 1 or feat_rows.stride(0) % 4 or feat_rows.data_ptr() % 16:
        feat_rows = feat_rows.contiguous()
    if label.dtype != torch.int64 or label.dim() != 3 or label.shape[0] != batch:
        raise ValueError("probe_train: label must be (batch, H, W) int64")
    label = label.contiguous()
    H, W = int(label.shape[1]), int(label.shape[2])
    dev = feat_rows.device
    lin = torch.empty((batch, H, W), dtype=torch.int64, device=dev)
    clu = torch.empty((batch, H, W), dtype=torch.int64, device=dev)
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    grad = torch.empty((batch * h * w, 2 * PROBE_SLOTS), dtype=torch.float32, device=dev)
    lib = load_library()
    nbytes = lib.equss_probe_workspace_bytes(batch, h, w, H, W)
    ws = _workspace(dev, nbytes, "probe")
    _check(lib.equss_probe_train(_ptr(feat_rows), feat_rows.stride(0), _ptr(probes), _ptr(bias), _ptr(label),
                                 _ptr(lin), _ptr(clu), _ptr(losses), _ptr(grad), _ptr(ws), ws.numel(),
                                 batch, h, w, feat_rows.shape[1], H, W, n_linear, n_cluster, _stream(feat_rows)),
           "equss_probe_train")
    return lin, clu, losses, grad


def confusion_update(confusion, preds, label, num_classes):
    """confusion (n_pred, num_classes) int64 on the GPU += counts of (pred, label) pairs with both in
    [0, num_classes)  (model/metric.py:45-58)."""
    _need_cuda(confusion, preds, label)
    if confusion.dtype != torch.int64 or not confusion.is_contiguous() or confusion.shape[1] != num_classes:
        raise ValueError("confusion_update: confusion must be a contiguous (n_pred, num_classes) int64 tensor")
    preds = preds.reshape(-1).long().contiguous()
    label = label.reshape(-1).long().contiguous()
    if preds.numel() != label.numel():
        raise ValueError("confusion_update: preds and label differ in size")
    _check(load_library().equss_confusion_update(_ptr(preds), _ptr(label), _ptr(confusion), preds.numel(),
                                                 int(num_classes), int(confusion.shape[0]), _stream(confusion)),
           "equss_confusion_update")


def index_rows_i16(z_index, label, upsample=True):
    """z_index (num_pq, B, h, w) int64, label (B, H, W) int64 on the GPU -> (rows, num_pq) int16 index table
    and (rows,) int16 labels (train.py:283-329): rows = B*H*W with ``upsample`` (indices nearest-upsampled
    to the label size) or B*h*w (labels sampled at the patch positions)."""
    _need_cuda(z_index, label)
    if z_index.dtype != torch.int64 or z_index.dim() != 4 or label.dtype != torch.int64 or label.dim() != 3:
        raise ValueError("index_rows_i16: z_index must be (num_pq, B, h, w) int64 and label (B, H, W) int64")
    num_pq, b, h, w = z_index.shape
    if label.shape[0] != b:
        raise ValueError("index_rows_i16: batch sizes of z_index and label differ")
    hh, ww = int(label.shape[1]), int(label.shape[2])
    z_index, label = z_index.contiguous(), label.contiguous()
    n = b * hh * ww if upsample else b * h * w
    idx_rows = torch.empty((n, num_pq), dtype=torch.int16, device=z_index.device)
    lab_rows = torch.empty((n,), dtype=torch.int16, device=z_index.device)
    _check(load_library().equss_index_rows_i16(_ptr(z_index), _ptr(label), _ptr(idx_rows), _ptr(lab_rows), b, h, w, hh, ww,
                                               num_pq, int(bool(upsample)), _stream(z_index)), "equss_index_rows_i16")
    return idx_rows, lab_rows


def code_usage(idx, num_codes, out=None):
    """idx (num_pq, rows) int64 -> counts (num_pq, num_codes) fp32 of how many rows chose each codeword
    (added into ``out`` when given)."""
    _need_cuda(idx)
    if idx.dtype != torch.int64 or idx.dim() != 2:
        raise ValueError("code_usage: idx must be (num_pq, rows) int64")
    idx = idx.contiguous()
    num_pq, rows = idx.shape
    if out is None:
        out = torch.zeros((num_pq, num_codes), dtype=torch.float32, device=idx.device)
    elif out.shape != (num_pq, num_codes) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("code_usage: out must be a contiguous (num_pq, num_codes) float32 tensor")
    _check(load_library().equss_code_usage(_ptr(idx), _ptr(out), rows, num_pq, int(num_codes), _stream(idx)), "equss_code_usage")
    return out


def code_sums(z, idx, num_codes, d):
    """z (rows, num_pq*d) fp32, idx (num_pq, rows) int64 -> sums (num_pq, num_codes, d) fp32: per codeword the sum of the
    rows assigned to it (the one_hot^T @ z of the EMA update, model/quantizer.py:485-491), all sub-spaces in one launch."""
    _need_cuda(z, idx)
    if idx.dtype != torch.int64 or idx.dim() != 2 or z.dtype != torch.float32 or z.dim() != 2 or z.stride(1) != 1:
        raise ValueError("code_sums: z must be (rows, num_pq*d) float32 with unit column stride, idx (num_pq, rows) int64")
    idx = idx.contiguous()
    num_pq, rows = idx.shape
    sums = torch.zeros((num_pq, num_codes, d), dtype=torch.float32, device=z.device)
    _check(load_library().equss_code_sums(_ptr(z), z.stride(0), _ptr(idx), _ptr(sums), rows, num_pq, int(num_codes), int(d),
                                          _stream(z)), "equss_code_sums")
    return sums
