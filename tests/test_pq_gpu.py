"""Parity of the HIP product-quantisation assign (through the C ABI) with the reference outputs
frozen in tests/golden and with the CPU oracle on fresh seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import assert_index_parity, regen_pq_case

pytestmark = pytest.mark.gpu

CASES = ["g4a_pq_s_random", "g6_pq_b_random", "g6b_pq_d16_k256", "g6c_pq_d32_k32", "g6d_pq_none_d32_k32",
         "g6e_pq_znorm"]


@pytest.mark.parametrize("name", CASES)
def test_pq_assign_matches_reference_fixture(name, native, golden, manifest, dev):
    g = golden(name)
    normalize = manifest["files"][name]["normalize"]
    rows, books = regen_pq_case(g["meta"], float(g["scale"]))
    idx, zq, loss = native.pq_assign(rows.to(dev), books.to(dev), normalize=normalize)
    torch.cuda.synchronize()
    assert idx.dtype == torch.int64 and tuple(idx.shape) == g["index"].shape
    flips = assert_index_parity(idx.cpu().numpy(), g["index"], g["gap64"], name)
    # z_q rows (skip rows whose index flipped on a near-tie)
    sel = g["sel"]
    same = (idx.cpu().numpy()[:, sel] == g["index"][:, sel]).all(axis=0)
    got = zq.cpu().numpy()[sel][same]
    np.testing.assert_allclose(got, g["zq_sel"][same], rtol=1e-5, atol=1e-6)
    # vq-loss = mean over sub-spaces of (book + beta) * mse   (dino_pqgo.py:685-687,:771-772)
    n, d = rows.shape[0], books.shape[2]
    vq = float((loss.cpu().double() / (n * d)).mean() * 1.25)
    assert abs(vq - float(g["vq_loss"])) <= 1e-5 * abs(float(g["vq_loss"])) + 1e-7, (vq, float(g["vq_loss"]), flips)


def test_pq_assign_on_pipeline_rows(native, golden, dev, pkg):
    g = golden("g4b_pq_pipeline_rows")
    state = pkg.synth.synth_pqgo_state(pkg.configs.S_CFG, seed=1234)
    books = torch.stack([state[f"model.vq_blocks.0.quantizers.{i}.embedding.weight"] for i in range(8)])
    rows = torch.from_numpy(g["code_rows"])
    idx, zq, loss = native.pq_assign(rows.to(dev), books.to(dev), normalize="l2")
    assert_index_parity(idx.cpu().numpy(), g["index"], g["gap64"], "g4b")
    same = (idx.cpu().numpy() == g["index"]).all(axis=0)
    np.testing.assert_allclose(zq.cpu().numpy()[same], g["zq_rows"][same], rtol=1e-5, atol=1e-6)
    vq = float((loss.cpu().double() / (rows.shape[0] * 128)).mean() * 1.25)
    assert abs(vq - float(g["vq_loss"])) <= 1e-5 * float(g["vq_loss"])
    prob = native.pq_prob(rows[:8].contiguous().to(dev), books.to(dev), "l2", tau=0.1)
    np.testing.assert_allclose(prob.cpu().numpy(), g["prob_rows"], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("rows_n", [1, 31, 64, 65, 1000])
def test_pq_assign_ragged_rows_vs_oracle(rows_n, native, oracle, dev):
    rng = np.random.Generator(np.random.PCG64(100 + rows_n))
    rows = torch.from_numpy(rng.standard_normal((rows_n, 1024)).astype(np.float32))
    books = torch.from_numpy(rng.uniform(-1, 1, size=(8, 256, 128)).astype(np.float32))
    idx, zq, loss = native.pq_assign(rows.to(dev), books.to(dev), "l2")
    want, gap = oracle.pq_top2_gap(list(books), rows, "l2")
    assert_index_parity(idx.cpu().numpy(), want.numpy(), gap.numpy(), f"rows={rows_n}")


def test_pq_assign_gather_normalized_and_strided_input(native, oracle, dev):
    """model/quantizer.py gathers the *normalised* codeword (:153,:474); input may be a column
    slice of a wider buffer."""
    rng = np.random.Generator(np.random.PCG64(7))
    wide = torch.from_numpy(rng.standard_normal((300, 96)).astype(np.float32)).to(dev)
    z = wide[:, :64]
    w = torch.from_numpy(rng.standard_normal((1, 128, 64)).astype(np.float32))
    idx, zq, loss = native.pq_assign(z, w.to(dev), "l2", gather_normalized=True)
    q, out, prob, oidx = oracle.ema_vector_quantizer_eval(w[0], z.cpu().contiguous(), "l2", 0.25, True)
    want, gap = oracle.pq_top2_gap([w[0]], z.cpu().contiguous(), "l2")
    assert_index_parity(idx.cpu().numpy(), want.numpy(), gap.numpy(), "ema")
    same = idx.cpu()[0] == oidx
    torch.testing.assert_close(zq.cpu()[same], q[same], rtol=1e-5, atol=1e-6)
    commit = float(loss.cpu()[0]) / z.numel()
    assert abs(0.25 * commit - float(out["loss"])) < 1e-5 * float(out["loss"]) + 1e-8


def test_pq_assign_rejects_bad_shapes(native, dev):
    z = torch.zeros(8, 96, device=dev)
    with pytest.raises(ValueError):
        native.pq_assign(z, torch.zeros(1, 32, 96, device=dev), "l2")       # d = 96 unsupported
    with pytest.raises(ValueError):
        native.pq_assign(torch.zeros(8, 64, device=dev), torch.zeros(1, 48, 64, device=dev), "l2")  # K % 32
    with pytest.raises(ValueError):
        native.pq_assign(torch.zeros(8, 64, device=dev), torch.zeros(1, 32, 64, device=dev), "bogus")
    idx, zq, loss = native.pq_assign(torch.zeros(0, 64, device=dev), torch.zeros(1, 32, 64, device=dev), "l2")
    assert idx.shape == (1, 0) and zq.shape == (0, 64)
