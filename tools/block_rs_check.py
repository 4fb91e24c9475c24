"""Developer tool: the register-streamed block-tail kernel (equss_dev_block_ws(3)) against the producer / consumer one (1)
inside the ViT forward: equal bits of the features at several batch sizes (ragged tiles included), and the time of the
block-tail launches either way (HIP-event pairs of the library's profiling hook)."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
if os.environ.get("EQUSS_LIB"):
    nat.LIB_PATH = os.environ["EQUSS_LIB"]
dev = torch.device("cuda", 0)
batches = [int(a) for a in sys.argv[1:]] or [3, 16, 64]
torch.manual_seed(0)
net = pkg.featurizer.vit_small(patch_size=8).to(dev).eval()
lib = nat.load_library()
lib.equss_dev_fuse_min_tiles(ctypes.c_int(1))
if os.environ.get("RS_STAGGER"):
    lib.equss_dev_rs_stagger(ctypes.c_int(int(os.environ["RS_STAGGER"])))
ok = True
for batch in batches:
    img = torch.randn(batch, 3, 320, 320, device=dev)
    outs, times = {}, {}
    for mode in (1, 3, 1, 3):
        lib.equss_dev_block_ws(ctypes.c_int(mode))
        with torch.no_grad():
            f = net.get_intermediate_feat(img, n=1)[0][0]
            torch.cuda.synchronize()
            nat.profile_enable(4096)
            for _ in range(3):
                net.get_intermediate_feat(img, n=1)
            prof = nat.profile_read(reset=True)
            nat.profile_enable(0)
        ms, n = prof["block_tail"]
        times[mode] = ms / max(n, 1) * 1e3
        outs.setdefault(mode, f.clone())
    eq = torch.equal(outs[1], outs[3])
    ok &= eq
    d = (outs[1] - outs[3]).abs().max().item()
    print(f"batch {batch:3d} rows {batch * 1601:6d}: block tail {times[1]:7.1f} us (ws) {times[3]:7.1f} us (rs)  equal bits {eq}  max|diff| {d:.3g} "
          f"finite {bool(torch.isfinite(outs[3]).all())}", flush=True)
lib.equss_dev_block_ws(ctypes.c_int(3))
lib.equss_dev_fuse_min_tiles(ctypes.c_int(384))
sys.exit(0 if ok else 1)
