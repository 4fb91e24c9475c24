"""Run one GEMM shape/variant a few times (for rocprofv3 counter passes)."""
import importlib, os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
lib = nat.load_library()
dev = torch.device("cuda", 0)
M = 64 * 1601
shapes = {"qkv": (M, 1152, 384, 0), "proj": (M, 384, 384, 4), "fc1": (M, 1536, 384, 1), "fc2": (M, 384, 1536, 4),
          "head2": (64 * 1600, 1024, 768, 3)}
name, variant = sys.argv[1], int(sys.argv[2])
m, n, k, epi = shapes[name]
STAGES = {62: 2 + 8, 63: 3 + 8, 64: 4 + 8, 32: 2, 33: 3, 34: 4, 205: 5 + 8, 206: 6 + 8, 207: 7 + 8}
if variant in STAGES:
    lib.equss_dev_gemm_variant(ctypes.c_int(3)); lib.equss_dev_gemm_stages(ctypes.c_int(STAGES[variant]))
else:
    lib.equss_dev_gemm_variant(ctypes.c_int(variant))
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
out = torch.zeros(m, n, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
for _ in range(5):
    nat.gemm_bf16(a, w, bias, epi, out=out)
torch.cuda.synchronize()
