"""Developer run: a few forwards of the fp32 parity mode at the benchmark shape (for rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
dev = torch.device("cuda", 0)
cfg = pkg.configs.pqgo_config()
m = pkg.build_model(cfg)
m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
m = m.to(dev).eval()
m.model.precision = "fp32"
img = pkg.synth.synth_images(64, 320, 320, seed=1).to(dev)
with torch.no_grad():
    for _ in range(5):
        m.model(img=img)
torch.cuda.synchronize()
