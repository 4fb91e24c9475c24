"""Run the attention and PQ kernels a few times at the benchmark shape (for rocprofv3 counter passes)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
if os.environ.get("EQUSS_LIB"):
    nat.LIB_PATH = os.environ["EQUSS_LIB"]
dev = torch.device("cuda", 0)
B, N, H = 64, 1601, 6
qkv = torch.randn(B * N, 3 * H * 64, device=dev).to(torch.bfloat16)
code = torch.randn(B * 1600, 1024, device=dev)
books = torch.rand(8, 256, 128, device=dev) - 0.5
for _ in range(4):
    out = nat.attention_bf16(qkv, B, N, H, 0.125)
    nat.pq_assign(code, books, "l2")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import ctypes
for diag in [int(a) for a in sys.argv[1:]] or [0]:
    nat.load_library().equss_dev_attn_diag(ctypes.c_int(diag))
    out = nat.attention_bf16(qkv, B, N, H, 0.125)
    if diag == 8 or diag == 0:
        ref = out.float().clone()
    else:
        print("   max |diff| vs 8 waves:", float((out.float() - ref).abs().max()))
    for _ in range(20):
        out = nat.attention_bf16(qkv, B, N, H, 0.125)
    e0.record()
    for _ in range(40):
        out = nat.attention_bf16(qkv, B, N, H, 0.125)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    print(f"attention diag {diag}: {us:.1f} us  {4.0 * B * N * N * 384 / us / 1e6:.1f} TF")
nat.load_library().equss_dev_attn_diag(ctypes.c_int(0))
if os.environ.get("ATTN_ONLY"):
    sys.exit(0)
for mode in (0, 1, 0):   # 0 = screened search where it applies, 1 = exhaustive fp32 search
    nat.load_library().equss_dev_pq_mode(ctypes.c_int(mode))
    nat.pq_assign(code, books, "l2")
    e0.record()
    for _ in range(10):
        nat.pq_assign(code, books, "l2")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"pq_assign mode {mode}: {us:.1f} us  {2.0 * B * 1600 * 1024 * 256 / us / 1e6:.1f} TF fp32  {B * 1600 * (8192 + 64) / us / 1e3:.1f} GB/s")
nat.load_library().equss_dev_pq_mode(ctypes.c_int(0))
