"""Developer micro-benchmark of equss_gemm_bf16 on the ViT-S/8 B=64 shapes (not part of the product).
usage: python tools/gemm_bench.py [variant ...]"""
import importlib, os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
if os.environ.get("EQUSS_LIB"):      # A/B against another build of the library
    nat.LIB_PATH = os.environ["EQUSS_LIB"]
lib = nat.load_library()
dev = torch.device("cuda", 0)
M = 64 * 1601
SHAPES = [("qkv", M, 1152, 384, 0), ("proj", M, 384, 384, 0), ("fc1", M, 1536, 384, 1), ("fc1_nogelu", M, 1536, 384, 0), ("fc2", M, 384, 1536, 0),
          ("head2", 64 * 1600, 1024, 768, 3)]
variants = [int(v) for v in sys.argv[1:]] or [0]
STAGES = {62: 2 + 8, 63: 3 + 8, 64: 4 + 8, 32: 2, 33: 3, 34: 4, 205: 5 + 8, 206: 6 + 8, 207: 7 + 8}
torch.manual_seed(0)
for name, m, n, k, epi in SHAPES:
    a = (torch.randn(m, k, device=dev)).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    out = torch.zeros(m, n, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    ref = None
    for v in variants:
        lib.equss_dev_ws_bk(ctypes.c_int(32 if v == 332 else 64))   # 332: the producer / consumer kernel with 32-wide K steps
        if v == 332:
            lib.equss_dev_gemm_variant(ctypes.c_int(3))
        elif v in STAGES:
            lib.equss_dev_gemm_variant(ctypes.c_int(3))
            lib.equss_dev_gemm_stages(ctypes.c_int(STAGES[v]))
        else:
            lib.equss_dev_gemm_variant(ctypes.c_int(v))
        out.zero_()
        nat.gemm_bf16(a, w, bias, epi, out=out)
        torch.cuda.synchronize()
        chk = out[:4096].float().clone()
        if ref is None:
            ref = chk
        err = float((chk - ref).abs().max())
        for _ in range(3):
            nat.gemm_bf16(a, w, bias, epi, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            nat.gemm_bf16(a, w, bias, epi, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:6s} variant {v}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF  (max diff vs variant {variants[0]}: {err:.3g})", flush=True)
