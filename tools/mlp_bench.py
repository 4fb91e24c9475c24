"""Developer micro-benchmark of the fused MLP kernel against the two-GEMM route (ViT-S/8, 64 x 1601 rows)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
lib = nat.load_library()
dev = torch.device("cuda", 0)
rows = [int(a) for a in sys.argv[1:]] or [64 * 1601, 32 * 1601, 768 * 128]
torch.manual_seed(0)
c, hid = 384, 1536
w1 = (torch.randn(hid, c, device=dev) / c ** 0.5).bfloat16(); b1 = torch.randn(hid, device=dev)
w2 = (torch.randn(c, hid, device=dev) / hid ** 0.5).bfloat16(); b2 = torch.randn(c, device=dev)
gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for m in rows:
    a = torch.randn(m, c, device=dev).bfloat16()
    x = torch.randn(m, c, device=dev)
    flops = 4.0 * m * c * hid
    us = timed(lambda: nat.mlp_resid_ln_bf16(a, w1, b1, w2, b2, x, gamma, beta))
    print(f"rows {m:7d}  {'fused':26s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)

    def two():
        h = nat.gemm_bf16(a, w1, b1, 1)
        nat.gemm_resid_ln_bf16(h, w2, b2, x, gamma, beta)
    us = timed(two)
    print(f"rows {m:7d}  {'two GEMM launches':26s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
