"""Developer tool: the producer / consumer block-tail kernel (equss_dev_block_ws) against the default one inside the ViT
forward: equal bits of the features, and the time of a forward either way."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
if os.environ.get("EQUSS_LIB"):
    nat.LIB_PATH = os.environ["EQUSS_LIB"]
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
net = pkg.featurizer.vit_small(patch_size=8).to(dev).eval()
img = torch.randn(batch, 3, 320, 320, device=dev)
lib = nat.load_library()
outs = {}
for on in (0, 1, 2, 0, 1, 2):
    lib.equss_dev_block_ws(ctypes.c_int(on))
    with torch.no_grad():
        f = net.get_intermediate_feat(img, n=1)[0][0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            net.get_intermediate_feat(img, n=1)
        e1.record(); torch.cuda.synchronize()
    print(f"block_ws={on}: {e0.elapsed_time(e1) / 5:.3f} ms per forward, finite {bool(torch.isfinite(f).all())}", flush=True)
    outs.setdefault(on, f.clone())
lib.equss_dev_block_ws(ctypes.c_int(3))
d = max((outs[0] - outs[1]).abs().max().item(), (outs[0] - outs[2]).abs().max().item())
print("equal bits:", torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]), " max |diff|:", d)
