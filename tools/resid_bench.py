"""Developer micro-benchmark of equss_gemm_resid_ln_bf16 (not part of the product).
usage: python tools/resid_bench.py"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
lib = nat.load_library()
dev = torch.device("cuda", 0)
M = 3 * 256 * 128
torch.manual_seed(0)
for name, k in (("proj", 384), ("fc2", 1536)):
    a = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(384, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(384, device=dev)
    gamma, beta = torch.ones(384, device=dev), torch.zeros(384, device=dev)
    x = torch.randn(M, 384, device=dev)
    for diag in [0]:
        for _ in range(3):
            nat.gemm_resid_ln_bf16(a, w, bias, x, gamma, beta, 1e-6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            nat.gemm_resid_ln_bf16(a, w, bias, x, gamma, beta, 1e-6)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name:5s} diag {diag:2d}: {us:8.1f} us  {2.0 * M * 384 * k / us / 1e6:7.1f} TF", flush=True)
        x.normal_()
lib.equss_dev_gemm_stages(ctypes.c_int(2 + 8))
