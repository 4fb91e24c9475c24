"""The model/quantizer.py family (VectorQuantizer :105-189, EMAVectorQuantizer :383-542 with its EMA
codebook :203-254, ProductQuantizerWrapper :587-611) running on the GPU through the HIP assign kernel,
against the reference outputs frozen in tests/golden/g8_quantizer_py.npz (eval mode) and
g8b_quantizer_train.npz (training mode: two forwards, so that the EMA buffers evolve)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = 2e-6     # fp32 values of size O(1) computed in a different summation order


def _hist(o):
    return [-1.0 if o[f"{a}-{b}"] is None else o[f"{a}-{b}"] for a in ("total", "current") for b in ("p10", "p50", "p90")]


def test_vector_quantizer_matches_reference_fixture(pkg, golden, dev):
    g = golden("g8_quantizer_py")
    z = torch.from_numpy(g["z"]).to(dev)
    vq = pkg.quantizer.VectorQuantizer(128, 64, beta=0.25, normalize="l2").to(dev).eval()
    with torch.no_grad():
        vq.codebook.weight.copy_(torch.from_numpy(g["weight"]))
        q, out, prob = vq(z)
    assert q.shape == z.shape and prob.shape == (120, 128)
    np.testing.assert_allclose(q.cpu().numpy(), g["vq_q"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(prob[:8].cpu().numpy(), g["vq_prob8"], rtol=1e-4, atol=1e-7)
    assert abs(float(out["loss"]) - float(g["vq_loss"])) < 1e-6
    assert abs(float(out["codebook_loss"]) - float(out["commitment_loss"])) < 1e-9
    # usage count of the call = bincount of the reference's indices (it counts in eval mode too, :157-164)
    want = np.bincount(g["vq_idx"].astype(np.int64), minlength=128)
    assert np.array_equal(vq.vq_count.cpu().numpy(), want)
    # with gradients: straight-through value identical, gradients reach z and the codebook
    z2 = z.clone().requires_grad_(True)
    q2, out2, _ = vq(z2)
    np.testing.assert_allclose(q2.detach().cpu().numpy(), g["vq_q"], rtol=0, atol=ATOL)
    out2["loss"].backward()
    assert z2.grad is not None and vq.codebook.weight.grad is not None
    assert float(vq.codebook.weight.grad.abs().sum()) > 0


def test_vector_quantizer_histograms_match_reference(pkg, golden, dev):
    g = golden("g8b_quantizer_train")
    k, d, _n = (int(v) for v in g["meta"])
    z4 = torch.from_numpy(g["z0"][:576]).reshape(2, 18, 16, d).permute(0, 3, 1, 2).contiguous().to(dev)
    vq = pkg.quantizer.VectorQuantizer(k, d, beta=0.25, normalize="l2").to(dev).eval()
    with torch.no_grad():
        vq.codebook.weight.copy_(torch.from_numpy(g["weight"]))
        for _ in range(2):
            q, out, _prob = vq(z4)
    assert np.array_equal(vq.vq_count.cpu().numpy(), g["vq_count"])
    assert _hist(out) == list(g["vq_hist"])
    np.testing.assert_allclose(q.cpu().numpy(), g["vq_q"], rtol=0, atol=ATOL)
    assert abs(float(out["loss"]) - float(g["vq_loss"])) < 1e-6


def test_ema_vector_quantizer_eval_and_training_match_reference(pkg, golden, dev):
    g8 = golden("g8_quantizer_py")
    zf = torch.from_numpy(g8["z"]).permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(dev)
    ema = pkg.quantizer.EMAVectorQuantizer(128, 64, beta=0.25, normalize="l2").to(dev).eval()
    ema.codebook.weight.copy_(torch.from_numpy(g8["weight"]))
    with torch.no_grad():
        q, out, prob = ema(zf)
    np.testing.assert_allclose(q.cpu().numpy(), g8["ema_q"], rtol=0, atol=ATOL)
    assert abs(float(out["loss"]) - float(g8["ema_loss"])) < 1e-6
    assert abs(float(out["codebook-sum"]) - float(g8["ema_codebook_sum"])) < 1e-4 * float(g8["ema_codebook_sum"])
    assert sorted(out.keys()) == ["codebook-sum", "commitment-loss", "loss"]

    g = golden("g8b_quantizer_train")
    k, d, _n = (int(v) for v in g["meta"])
    ema = pkg.quantizer.EMAVectorQuantizer(k, d, beta=0.25, normalize="l2", decay=float(g["decay"]),
                                           eps=float(g["eps"])).to(dev).train()
    w = torch.from_numpy(g["weight"])
    ema.codebook.weight.copy_(w)
    ema.codebook.weight_avg.copy_(w)
    for step in range(2):
        z = torch.from_numpy(g[f"z{step}"]).to(dev)
        q, out, prob = ema(z)
        np.testing.assert_allclose(q.cpu().numpy(), g[f"s{step}_q"], rtol=0, atol=ATOL)
        assert abs(float(out["loss"]) - float(g[f"s{step}_loss"])) < 1e-6
        assert abs(out["codebook-usage"] - float(g[f"s{step}_usage"])) < 1e-7
        assert _hist(out) == list(g[f"s{step}_hist"])
        assert np.array_equal(ema.vq_count.cpu().numpy(), g[f"s{step}_vq_count"])
        np.testing.assert_allclose(ema.codebook.vq_count.cpu().numpy(), g[f"s{step}_ema_count"], rtol=1e-6)
        np.testing.assert_allclose(ema.codebook.weight_avg.cpu().numpy(), g[f"s{step}_weight_avg"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ema.codebook.weight.cpu().numpy(), g[f"s{step}_weight"], rtol=1e-4, atol=1e-5)
        assert abs(float(out["codebook-sum"]) - float(g[f"s{step}_codebook_sum"])) < 1e-4 * float(g[f"s{step}_codebook_sum"])


def test_quantizer_py_product_wrapper_matches_reference(pkg, golden, dev):
    g = golden("g8_quantizer_py")
    zf = torch.from_numpy(g["z"]).permute(0, 2, 3, 1).reshape(-1, 64).contiguous().to(dev)
    pqw = pkg.quantizer.ProductQuantizerWrapper(4, 32, 64, beta=0.25, normalize="l2").to(dev).eval()
    for i, b in enumerate(g["pq_books"]):
        pqw.quantizers[i].codebook.weight.copy_(torch.from_numpy(b))
    with torch.no_grad():
        q, out, prob = pqw(zf)
    assert q.shape == (120, 64) and prob.shape == (120, 4 * 32)
    np.testing.assert_allclose(q.cpu().numpy(), g["pq_q"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(prob[:8].cpu().numpy(), g["pq_prob8"], rtol=1e-4, atol=1e-7)
    assert abs(float(out["loss"]) - float(g["pq_loss"])) < 1e-6
    # VectorQuantizer rejects the decay / eps arguments this wrapper passes, as in the reference (:579-585)
    with pytest.raises(TypeError):
        pkg.quantizer.ProductQuantizerWrapper(4, 32, 64, quantizer_cls=pkg.quantizer.VectorQuantizer)


def test_product_wrapper_batched_path_equals_the_loop_over_quantisers(pkg, dev):
    """quantizer.py:587-611 loops over the sub-quantisers; the wrapper here folds that loop into one assign launch, one
    probability launch, one usage histogram, one scatter-add and one all-reduce.  Same outputs and the same EMA buffers
    after two training steps and in eval mode (sums of a codeword's rows come in a different order: rounding only)."""
    import copy
    torch.manual_seed(11)
    a = pkg.quantizer.ProductQuantizerWrapper(4, 64, 128, beta=0.25, normalize="l2", decay=0.9).to(dev).train()
    for q in a.quantizers:
        q.codebook.weight.normal_()
        q.codebook.weight_avg.copy_(q.codebook.weight)
    b = copy.deepcopy(a)
    b.batched = False
    for step in range(3):
        if step == 2:
            a.eval(); b.eval()
        z = torch.randn(1000, 128, device=dev)
        with torch.no_grad():
            qa, oa, pa = a(z)
            qb, ob, pb = b(z)
        assert torch.allclose(qa, qb, rtol=0, atol=ATOL) and torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
        assert sorted(oa.keys()) == sorted(ob.keys())
        for k in oa:
            assert abs(float(oa[k]) - float(ob[k])) <= 1e-6 + 1e-5 * abs(float(ob[k])), k
        for x, y in zip(a.quantizers, b.quantizers):
            assert torch.equal(x.vq_count, y.vq_count)
            assert torch.allclose(x.codebook.vq_count, y.codebook.vq_count, rtol=1e-6)
            assert torch.allclose(x.codebook.weight_avg, y.codebook.weight_avg, rtol=1e-5, atol=1e-6)
            assert torch.allclose(x.codebook.weight, y.codebook.weight, rtol=1e-4, atol=1e-5)


def test_code_sums_match_the_one_hot_product(native, dev):
    """equss_code_sums against one_hot^T @ z (model/quantizer.py:485-491), two sub-spaces, a codeword nobody chose."""
    torch.manual_seed(5)
    z = torch.randn(777, 64, device=dev)
    idx = torch.randint(0, 15, (2, 777), device=dev)   # codeword 15 stays unused
    got = native.code_sums(z, idx, 16, 32)
    for p in range(2):
        want = torch.nn.functional.one_hot(idx[p], 16).float().t() @ z[:, p * 32:(p + 1) * 32]
        assert torch.allclose(got[p], want, rtol=1e-5, atol=1e-5)
    assert float(got[:, 15].abs().max()) == 0.0
