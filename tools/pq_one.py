"""Run the PQ assign kernel a few times at the benchmark shape (for rocprofv3 counter passes and timing).
usage: python tools/pq_one.py [cfg: s|b] [repeats]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
if os.environ.get("EQUSS_LIB"):      # A/B against another build of the library
    nat.LIB_PATH = os.environ["EQUSS_LIB"]
dev = torch.device("cuda", 0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "s"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
if cfg == "s":
    rows, num_pq, k = 64 * 1600, 8, 256
else:
    rows, num_pq, k = 4 * 9216, 16, 512
if os.environ.get("PQ_MODE"):
    import ctypes
    nat.load_library().equss_dev_pq_mode(ctypes.c_int(int(os.environ["PQ_MODE"])))
code = torch.randn(rows, num_pq * 128, device=dev)
books = torch.rand(num_pq, k, 128, device=dev) - 0.5
for _ in range(3):
    nat.pq_assign(code, books, "l2")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    idx, zq, loss = nat.pq_assign(code, books, "l2")
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"pq_assign {cfg}-cfg rows={rows} num_pq={num_pq} K={k}: {us:.1f} us  {rows * num_pq * (1024 + 8) / us / 1e3:.1f} GB/s  checksum {int(idx.sum())}")
