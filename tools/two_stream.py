"""Developer experiment: the 64-image forward as two 32-image halves on two HIP streams, each launch
limited to a share of the CUs, against the single-stream 64-image forward.
usage: python tools/two_stream.py [cu_share ...]"""
import ctypes, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
lib = pkg.native.load_library()
dev = torch.device("cuda", 0)
cfg = pkg.configs.S_CFG
m = pkg.build_model(cfg)
m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
m = m.to(dev).eval()
img = pkg.synth.synth_images(64, 320, 320, seed=1).to(dev)
def split(k):
    n = 64 // k
    return [img[i * n:(i + 1) * n].contiguous() for i in range(k)], [torch.cuda.Stream() for _ in range(k)]


halves, streams = split(2)


def one(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            m.model(img=img)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def two(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(steps):
            for s, h in zip(streams, halves):
                with torch.cuda.stream(s):
                    m.model(img=h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one(3)
print(f"single stream, 64 images: {one(10):.3f} ms", flush=True)
for share in [int(a) for a in sys.argv[1:]] or [256, 128, 160]:
    lib.equss_dev_gemm_cu_share(ctypes.c_int(share))
    two(3)
    print(f"two streams x 32 images, GEMM launches on {share} CUs: {two(10):.3f} ms", flush=True)
lib.equss_dev_gemm_cu_share(ctypes.c_int(256))
for k in (2, 4, 8):
    halves, streams = split(k)
    two(3)
    print(f"{k} streams x {64 // k} images: {two(10):.3f} ms", flush=True)
print(f"single stream again: {one(10):.3f} ms")
