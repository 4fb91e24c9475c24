"""Developer tool: where a tile of the register-streamed block-tail kernel spends its time.  Needs a stamp build of the library
(tools/build_ablations.sh RSSTAMPS [+ other RS ablations]) named by EQUSS_LIB; prints, per tile of workgroup 77 in the last
block-tail launch of a ViT forward, the shader-clock cycles between phase boundaries (mean and max over the eight waves)."""
import ctypes, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
nat = pkg.native
nat.LIB_PATH = os.environ["EQUSS_LIB"]
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
net = pkg.featurizer.vit_small(patch_size=8).to(dev).eval()
lib = nat.load_library()
lib.equss_dev_block_ws(ctypes.c_int(3))
if os.environ.get("RS_STAGGER"):
    lib.equss_dev_rs_stagger(ctypes.c_int(int(os.environ["RS_STAGGER"])))
img = torch.randn(batch, 3, 320, 320, device=dev)
with torch.no_grad():
    for _ in range(3):
        net.get_intermediate_feat(img, n=1)
torch.cuda.synchronize()
NW, MAXS = 8, 160
buf = (ctypes.c_uint32 * (NW * MAXS))()
rc = lib.equss_dev_rs_stamps(buf)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint32).reshape(NW, MAXS).astype(np.int64)
LN = ["sums a", "wait", "sums b", "wait"]
HEIGHTS = [int(h) for h in os.environ.get("RS_HEIGHTS", "6,7,6,6").split(",")]   # row groups of workgroup 77's tiles


def tile_names(ng):
    names = ["rows staged", "G0"] + ["LN2 " + x for x in LN] + ["LN2 normalise"]
    for c in range(6):
        names += [f"c{c} G1", f"c{c} wait hidden free", f"c{c} GELU", f"c{c} wait hidden done", f"c{c} G2"]
    names += ["(last G2 stamp)", "wait hidden free"] + ["LN " + x for x in LN]
    for r in range(ng):
        names += [f"out{r} normalise", f"out{r} wait", f"out{r} copy + stores", f"out{r} wait free"]
    return names + ["tail"]


tot = {}
pos = 0
print("cycles (shader clock) mean / max over waves")
for t, ng in enumerate(HEIGHTS):
    names = tile_names(ng)
    if pos + len(names) + 1 > MAXS or st[0, pos + len(names)] == 0:
        break
    seg = st[:, pos:pos + len(names) + 1]
    d = (np.diff(seg, axis=1) & 0xffffffff)
    print(f"tile {t} ({ng} groups): total {d.sum(axis=1).mean():9.0f}")
    for i, nm in enumerate(names):
        key = nm[3:] if nm[0] == "c" and nm[1].isdigit() else (nm[5:] if nm.startswith("out") else nm)
        key = "out " + key if nm.startswith("out") else key
        tot.setdefault(key, []).append(d[:, i].mean())
        if t == int(os.environ.get("RS_SHOW", "1")):
            print(f"   {nm:24s} {d[:, i].mean():8.0f} {d[:, i].max():8.0f}")
    pos += len(names) + 1
print("sum over the launch by phase (cycles per workgroup, mean over waves):")
for k, v in tot.items():
    print(f"   {k:24s} {sum(v):9.0f}")
