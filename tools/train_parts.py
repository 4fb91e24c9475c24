"""Developer timing of the torch-side pieces of the training step at the benchmark shape (64 x 320x320):
the PQ straight-through backward, the head's backward and the probes' loss with its backward."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H, W, D, C = 64, 40, 40, 1024, 384
n = B * H * W


def timed(name, fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / reps:8.3f} ms")


rows = torch.randn(n, D, device=dev, requires_grad=True)
books = ((torch.rand(8, 256, 128, device=dev) - 0.5) / 128).requires_grad_(True)
g = torch.randn(n, D, device=dev) * 1e-3


def pq_fb():
    zq, vq, idx = pkg.pqgo._PQStraightThrough.apply(rows, books, 0.25, 1.0)
    torch.autograd.grad([zq, vq], [rows, books], [g, torch.ones_like(vq)])


def pq_f():
    with torch.no_grad():
        pkg.pqgo._PQStraightThrough.apply(rows, books, 0.25, 1.0)


timed("pq forward", pq_f)
timed("pq forward + backward", pq_fb)

feat = torch.randn(n, C, device=dev)
ps = [torch.randn(D, C, 1, 1, device=dev) * 0.05, torch.randn(D, device=dev) * 0.1,
      torch.randn(C, C, 1, 1, device=dev) * 0.05, torch.randn(C, device=dev) * 0.1,
      torch.randn(D, C, 1, 1, device=dev) * 0.05, torch.randn(D, device=dev) * 0.1]
ps = [p.requires_grad_(True) for p in ps]


def head_f():
    with torch.no_grad():
        pkg.pqgo._HeadFunction.apply(feat, *ps)


def head_fb():
    code = pkg.pqgo._HeadFunction.apply(feat, *ps)
    torch.autograd.grad(code, ps, g)


timed("head forward", head_f)
timed("head forward + backward", head_fb)

ev = pkg.evaluator.UnSegEvaluator(D, 27).to(dev).train()
out = torch.randn(B, H, W, D, device=dev).permute(0, 3, 1, 2)
label = torch.randint(-1, 27, (B, 320, 320), device=dev)
img = None


def probes_fb():
    ll, _, cl, _ = ev(out, img, label)
    torch.autograd.grad(ll + cl, list(ev.parameters()))


def probes_f():
    with torch.no_grad():
        ev(out, img, label)


timed("probes forward (no grad, HIP)", probes_f)
timed("probes forward + backward", probes_fb)

nat = pkg.native
rows_p = out.permute(0, 2, 3, 1).reshape(B * H * W, D)
probes_m, bias_m = nat.pack_probes(ev.linear_probe.weight.detach(), ev.linear_probe.bias.detach(), ev.cluster_probe.clusters.detach())
timed("  pack_probes", lambda: nat.pack_probes(ev.linear_probe.weight.detach(), ev.linear_probe.bias.detach(), ev.cluster_probe.clusters.detach()))
timed("  probe_eval", lambda: nat.probe_eval(rows_p, probes_m, bias_m, label, B, H, W, 27, 27))
timed("  probe_train", lambda: nat.probe_train(rows_p, probes_m, bias_m, label, B, H, W, 27, 27))
gtab = nat.probe_train(rows_p, probes_m, bias_m, label, B, H, W, 27, 27)[3]
timed("  grad^T x rows (torch.mm)", lambda: torch.mm(gtab.t(), rows_p))

g_bf = g.to(torch.bfloat16)
fh = torch.randn(n, 2 * C, device=dev).to(torch.bfloat16)
w_cat = torch.randn(D, 2 * C, device=dev).to(torch.bfloat16)
timed("  g.to(bf16)", lambda: g.to(torch.bfloat16))
timed("  gemm_tn (1024 x 768)", lambda: nat.gemm_tn_bf16(g_bf, fh))
timed("  torch.mm (1024 x 768)", lambda: torch.mm(g_bf.t(), fh))
w2bt = w_cat[:, C:].t().contiguous()
timed("  gemm_bf16 d_pre", lambda: nat.gemm_bf16(g_bf, w2bt, None, nat.EPI_BF16))
d_pre = nat.gemm_bf16(g_bf, w2bt, None, nat.EPI_BF16)
timed("  relu mask", lambda: d_pre * (fh[:, C:] > 0))
timed("  gemm_tn (384 x 384)", lambda: nat.gemm_tn_bf16(d_pre, fh[:, :C]))
timed("  g.sum(0)", lambda: g.sum(0))
timed("  d_pre.float().sum(0)", lambda: d_pre.float().sum(0))
