"""Parity of the HIP probe path (equss_probe_eval through the C ABI, reached from UnSegEvaluator)
with the reference's evaluator outputs frozen in tests/golden/g9_evaluator.npz, and of the metric
accumulation on the device with g10_metrics.npz."""
import numpy as np
import pytest
import torch

from conftest import assert_index_parity, regen_probe_case

pytestmark = pytest.mark.gpu

# class scores are O(1); fp32 evaluation orders differ by ~1e-6 of that
LOGIT_TIE = 2e-5
COSINE_TIE = 2e-6


def _evaluator(pkg, sd, d, ncls, dev):
    ev = pkg.UnSegEvaluator(d, ncls, 0)
    ev.load_state_dict({k[len("evaluator."):]: v for k, v in sd.items()}, strict=True)
    return ev.to(dev).eval()


@pytest.mark.parametrize("name", ["x8", "frac", "same"])
def test_probe_eval_matches_reference_fixture(name, pkg, golden, dev):
    g = golden("g9_evaluator")
    out, label, sd = regen_probe_case(g[f"{name}_meta"], g[f"{name}_seed"])
    b, d, h, w, hh, ww, ncls = (int(v) for v in g[f"{name}_meta"])
    ev = _evaluator(pkg, sd, d, ncls, dev)
    # token-major storage with a channels-first view, as the model hands it over
    rows = out.permute(0, 2, 3, 1).contiguous().to(dev)
    view = rows.permute(0, 3, 1, 2)
    with torch.no_grad():
        lin_loss, lin_p, clu_loss, clu_p = ev(view, None, label=label.to(dev))
    torch.cuda.synchronize()
    assert lin_p.dtype == torch.int64 and tuple(lin_p.shape) == (b, hh, ww) == tuple(clu_p.shape)
    assert_index_parity(lin_p.cpu().numpy(), g[f"{name}_linear_preds"], g[f"{name}_linear_gap64"], name + " linear", LOGIT_TIE)
    assert_index_parity(clu_p.cpu().numpy(), g[f"{name}_cluster_preds"], g[f"{name}_cluster_gap64"], name + " cluster", COSINE_TIE)
    assert abs(float(lin_loss) - float(g[f"{name}_linear_loss"])) <= 2e-5 * abs(float(g[f"{name}_linear_loss"])) + 1e-6
    assert abs(float(clu_loss) - float(g[f"{name}_cluster_loss"])) <= 2e-5 * abs(float(g[f"{name}_cluster_loss"])) + 1e-6


def test_probe_eval_contiguous_nchw_input_and_stock_path_agree(pkg, golden, dev):
    """A plain NCHW tensor is accepted (copied to token-major), and the autograd path used while the
    probes are trained gives the same predictions."""
    g = golden("g9_evaluator")
    out, label, sd = regen_probe_case(g["frac_meta"], g["frac_seed"])
    b, d, h, w, hh, ww, ncls = (int(v) for v in g["frac_meta"])
    ev = _evaluator(pkg, sd, d, ncls, dev)
    with torch.no_grad():
        hip = ev(out.to(dev), None, label=label.to(dev))
    ev.train()
    stock = ev(out.to(dev), None, label=label.to(dev))   # grad enabled -> stock ops
    assert stock[0].requires_grad and stock[2].requires_grad
    assert_index_parity(hip[1].cpu().numpy(), stock[1].cpu().numpy(), g["frac_linear_gap64"], "linear", LOGIT_TIE)
    assert_index_parity(hip[3].cpu().numpy(), stock[3].cpu().numpy(), g["frac_cluster_gap64"], "cluster", COSINE_TIE)
    assert torch.allclose(hip[0], stock[0].detach(), rtol=2e-5) and torch.allclose(hip[2], stock[2].detach(), rtol=2e-5)


@pytest.mark.parametrize("name", ["x8", "frac", "same", "down"])
def test_probe_training_gradients_match_autograd(name, pkg, golden, dev):
    """The probes' closed-form gradients from the HIP kernels (equss_probe_train: gradient table of the class maps, then
    table^T x features) against torch autograd through the stock-op restatement of evaluator.py:46-82 (up-sampled class
    maps, log-sum-exp cross-entropy, max cosine): losses, predictions and the gradients of linear_probe.weight / .bias and
    cluster_probe.clusters.  'down' (label smaller than the feature map) takes the kernel's path without an LDS window."""
    g = golden("g9_evaluator")
    if name == "down":
        b, d, h, w, hh, ww, ncls = 2, 64, 16, 12, 7, 5, 27
        gen = torch.Generator(device="cpu").manual_seed(11)
        out = torch.randn(b, d, h, w, generator=gen)
        label = torch.randint(-1, ncls, (b, hh, ww), generator=gen)
        ev = pkg.UnSegEvaluator(d, ncls, 0).to(dev)
    else:
        out, label, sd = regen_probe_case(g[f"{name}_meta"], g[f"{name}_seed"])
        b, d, h, w, hh, ww, ncls = (int(v) for v in g[f"{name}_meta"])
        ev = _evaluator(pkg, sd, d, ncls, dev)
    ev.train()
    params = [ev.linear_probe.weight, ev.linear_probe.bias, ev.cluster_probe.clusters]
    res = {}
    for hip in (True, False):
        ev.hip_grads = hip
        lin_loss, lin_p, clu_loss, clu_p = ev(out.to(dev), None, label=label.to(dev))
        grads = torch.autograd.grad(0.7 * lin_loss + 1.3 * clu_loss, params)
        res[hip] = (lin_loss.detach(), clu_loss.detach(), lin_p, clu_p, grads)
    torch.cuda.synchronize()
    a, r = res[True], res[False]
    assert torch.allclose(a[0], r[0], rtol=2e-5) and torch.allclose(a[1], r[1], rtol=2e-5)
    assert float((a[2] != r[2]).float().mean()) < 1e-3 and float((a[3] != r[3]).float().mean()) < 1e-3   # near-ties only
    for pname, ga, gr in zip(("weight", "bias", "clusters"), a[4], r[4]):
        assert ga.shape == gr.shape, pname
        err = float((ga - gr).norm() / gr.norm())
        print(name, pname, "gradient rel-L2 error", err)
        # (a pixel whose two best cosines tie to ~1e-7 may pick the other class in the two evaluation orders)
        assert err < (2e-3 if pname == "clusters" else 2e-5), (pname, err)


def test_probe_eval_no_labelled_pixel_gives_nan_loss(pkg, golden, dev):
    g = golden("g9_evaluator")
    out, label, sd = regen_probe_case(g["same_meta"], g["same_seed"])
    b, d, h, w, hh, ww, ncls = (int(v) for v in g["same_meta"])
    ev = _evaluator(pkg, sd, d, ncls, dev)
    with torch.no_grad():
        lin_loss, _, clu_loss, _ = ev(out.to(dev), None, label=torch.full_like(label, -1).to(dev))
    assert torch.isnan(lin_loss) and torch.isfinite(clu_loss)   # mean over an empty selection, evaluator.py:78-80


def test_probe_eval_argument_errors(native, dev):
    feat = torch.zeros(4 * 4, 60, device=dev)
    probes = torch.zeros(64, 60, device=dev)
    bias = torch.zeros(64, device=dev)
    label = torch.zeros(1, 8, 8, dtype=torch.int64, device=dev)
    with pytest.raises(ValueError):      # D % 8 != 0
        native.probe_eval(feat, probes, bias, label, 1, 4, 4, 5, 5)
    feat = torch.zeros(4 * 4, 64, device=dev)
    probes = torch.zeros(64, 64, device=dev)
    with pytest.raises(ValueError):      # more classes than table slots
        native.probe_eval(feat, probes, bias, label, 1, 4, 4, 33, 5)


def test_full_scale_probe_properties(pkg, native, dev):
    """At the benchmark size (64 x 320 x 320 pixels) no oracle finishes in seconds; check the
    properties the computation must have: predictions constant where the low-res field is constant,
    a permutation of the class order permutes the predictions, and both losses are reproducible."""
    b, h, w, d, hh, ww, ncls = 64, 40, 40, 1024, 320, 320, 27
    gen = torch.Generator(device="cpu").manual_seed(7)
    cell = torch.randn(b, 1, 1, d, generator=gen).expand(b, h, w, d).contiguous()   # constant per image
    rows = cell.reshape(-1, d).to(dev)
    weight = (0.05 * torch.randn(ncls, d, generator=gen)).to(dev)
    biasv = (0.1 * torch.randn(ncls, generator=gen)).to(dev)
    clusters = torch.randn(ncls, d, generator=gen).to(dev)
    label = torch.randint(-1, ncls, (b, hh, ww), generator=gen).to(dev)
    probes, bias = native.pack_probes(weight, biasv, clusters)
    lin, clu, losses = native.probe_eval(rows, probes, bias, label, b, h, w, ncls, ncls)
    want_lin = (rows[:: h * w] @ weight.t() + biasv).argmax(1)
    want_clu = (torch.nn.functional.normalize(rows[:: h * w], dim=1) @ torch.nn.functional.normalize(clusters, dim=1).t()).argmax(1)
    assert torch.equal(lin, want_lin.view(b, 1, 1).expand(b, hh, ww))
    assert torch.equal(clu, want_clu.view(b, 1, 1).expand(b, hh, ww))
    # cluster loss of a constant field = -mean of the per-image max cosine
    cos = (torch.nn.functional.normalize(rows[:: h * w], dim=1) @ torch.nn.functional.normalize(clusters, dim=1).t()).max(1).values
    assert abs(float(losses[1]) + float(cos.mean())) < 1e-5
    perm = torch.randperm(ncls, generator=gen).to(dev)
    probes_p, bias_p = native.pack_probes(weight[perm], biasv[perm], clusters[perm])
    lin_p, clu_p, losses_p = native.probe_eval(rows, probes_p, bias_p, label, b, h, w, ncls, ncls)
    assert torch.equal(perm[lin_p], lin) and torch.equal(perm[clu_p], clu)
    lin2, clu2, losses2 = native.probe_eval(rows, probes, bias, label, b, h, w, ncls, ncls)
    assert torch.equal(losses, losses2) and torch.equal(lin, lin2) and torch.equal(clu, clu2)


@pytest.mark.parametrize("name", ["cluster", "linear", "extra", "extra_mid"])
def test_metrics_on_device_match_reference_fixture(name, pkg, golden, dev):
    from importlib import import_module
    metric = import_module(pkg.__name__ + ".metric")
    g = golden("g10_metrics")
    ncls, extra, hung = (int(v) for v in g[f"{name}_meta"])
    m = metric.UnSegMetrics(ncls, extra, bool(hung), dev)
    preds = torch.from_numpy(g[f"{name}_preds"].astype(np.int64)).to(dev)
    label = torch.from_numpy(g[f"{name}_label"].astype(np.int64)).to(dev)
    m.update(preds[:1], label[:1])
    m.update(preds[1:], label[1:])
    res = m.compute(name)
    assert np.array_equal(m.confusion_matrix.cpu().numpy(), g[f"{name}_confusion"])
    assert abs(float(res["iou"]) - float(g[f"{name}_iou"])) < 1e-4
    assert abs(float(res["accuracy"]) - float(g[f"{name}_accuracy"])) < 1e-4


def test_index_dump_rows_and_usage_counts_on_device(pkg, native, dev, tmp_path):
    """The index-dump kernel (int16 rows for the reference's offline analyses, train.py:283-329) against
    F.interpolate(mode="nearest") on the index maps, both sampling directions, odd ratios and num_pq with and
    without a specialised instance; CodeIndexDump with device tensors against the same with host tensors; the
    usage-count kernel against a bincount per sub-space; get_histogram_count on a device tensor."""
    import torch.nn.functional as F
    gen = torch.Generator(device="cpu").manual_seed(4)
    for num_pq, b, h, w, hh, ww in ((8, 3, 5, 4, 40, 32), (3, 2, 7, 5, 45, 33), (16, 1, 6, 6, 6, 6)):
        z = torch.randint(0, 512, (num_pq, b, h, w), generator=gen)
        lab = torch.randint(-1, 27, (b, hh, ww), generator=gen)
        rows, labs = native.index_rows_i16(z.to(dev), lab.to(dev), upsample=True)
        up = F.interpolate(z.permute(1, 0, 2, 3).float(), size=(hh, ww), mode="nearest")
        want = up.reshape(b, num_pq, -1).permute(0, 2, 1).reshape(-1, num_pq).to(torch.int16)
        assert rows.dtype == torch.int16 and torch.equal(rows.cpu(), want)
        assert torch.equal(labs.cpu(), lab.reshape(-1).to(torch.int16))
        rows2, labs2 = native.index_rows_i16(z.to(dev), lab.to(dev), upsample=False)
        assert torch.equal(rows2.cpu(), z.permute(1, 2, 3, 0).reshape(-1, num_pq).to(torch.int16))
        down = F.interpolate(lab[:, None].float(), size=(h, w), mode="nearest")[:, 0]
        assert torch.equal(labs2.cpu(), down.reshape(-1).to(torch.int16))
    # CodeIndexDump: device path == host path, file by file
    z = torch.randint(0, 256, (8, 2, 8, 8), generator=gen)
    lab = torch.randint(-1, 27, (2, 64, 64), generator=gen)
    for name, zz, ll in (("dev", z.to(dev), lab.to(dev)), ("host", z, lab)):
        with pkg.CodeIndexDump(str(tmp_path / name), 8) as dump:
            dump.append(zz, ll)
            dump.append(zz[:, :1], ll[:1])
    for suffix in ("_index.npy", "_label.npy"):
        a, bb = np.load(str(tmp_path / "dev") + suffix), np.load(str(tmp_path / "host") + suffix)
        assert a.shape == bb.shape and a.dtype == np.int16 and np.array_equal(a, bb)
    # usage counts
    idx = torch.randint(0, 256, (8, 100000), generator=gen)
    idx[3] = 7
    counts = native.code_usage(idx.to(dev), 256)
    want = torch.stack([torch.bincount(idx[p], minlength=256) for p in range(8)]).float()
    assert torch.equal(counts.cpu(), want)
    native.code_usage(idx.to(dev)[:, :10], 256, out=counts)
    assert float(counts.sum()) == 8 * 100010
    hist = pkg.quantizer.get_histogram_count(counts[0], "current")
    assert hist == pkg.quantizer.get_histogram_count(counts[0].cpu(), "current")
    assert pkg.quantizer.get_histogram_count(counts[3], "t") == {"t-p10": 0.0, "t-p50": 0.0, "t-p90": 0.0}
