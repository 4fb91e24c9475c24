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


# ---- the screened search (fp16 MFMA scores + exact fp32 re-scoring of the survivors) -------------------
def _force_exact(native, on):
    import ctypes
    native.load_library().equss_dev_pq_mode(ctypes.c_int(1 if on else 0))


def _check_against_fp64(idx, want, gap, what, tol=2e-6):
    got = idx.cpu().numpy()
    bad = (got != want.numpy()) & (gap.numpy() > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())} mismatches outside near-ties"


def _adversarial_case(kind, rows, k, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    d, num_pq = 128, 2
    code = g.standard_normal((rows, num_pq * d)).astype(np.float32)
    books = g.uniform(-1, 1, size=(num_pq, k, d)).astype(np.float32)
    if kind == "clustered":       # codewords in tight clusters: many survivors per row, lists overflow
        centres = g.standard_normal((num_pq, 4, d)).astype(np.float32)
        books = centres[:, g.integers(0, 4, size=k)] + 1e-3 * g.standard_normal((num_pq, k, d)).astype(np.float32)
    elif kind == "duplicates":    # exact duplicate codewords: exact ties, the lower index must win
        books[:, k // 2:] = books[:, : k - k // 2]
    elif kind == "on_codeword":   # rows that ARE codewords (distance 0 up to rounding), zero and tiny rows
        for p in range(num_pq):
            pick = g.integers(0, k, size=rows)
            code[:, p * d:(p + 1) * d] = 3.0 * books[p, pick]
        code[::7] = 0.0
        code[3::11] *= 1e-20
    return torch.from_numpy(code), torch.from_numpy(books)


@pytest.mark.parametrize("kind,rows,k", [("random", 1000, 256), ("random", 333, 128), ("random", 777, 64),
                                         ("random", 500, 512), ("clustered", 300, 512),
                                         ("clustered", 600, 256), ("duplicates", 500, 128),
                                         ("on_codeword", 900, 256)])
def test_screened_search_equals_exact_search(kind, rows, k, native, oracle, dev):
    code, books = _adversarial_case(kind, rows, k, seed=rows + k)
    want, gap = oracle.pq_top2_gap([b for b in books], code, "l2")
    try:
        _force_exact(native, True)
        idx_e, zq_e, loss_e = native.pq_assign(code.to(dev), books.to(dev))
        _force_exact(native, False)
        idx_s, zq_s, loss_s = native.pq_assign(code.to(dev), books.to(dev))
    finally:
        _force_exact(native, False)
    torch.cuda.synchronize()
    _check_against_fp64(idx_e, want, gap, kind + " exact kernel")
    _check_against_fp64(idx_s, want, gap, kind + " screened kernel")
    if kind == "duplicates":      # ties between identical codewords resolve to the first one in both kernels
        assert int(idx_s.max()) < k - k // 2 + (k % 2) and torch.equal(idx_s, idx_e)
    same = (idx_s == idx_e).all(dim=0).cpu().numpy()
    np.testing.assert_allclose(zq_s.cpu().numpy()[same], zq_e.cpu().numpy()[same], rtol=0, atol=1e-6)
    assert torch.allclose(loss_s, loss_e, rtol=1e-4)
    # the screened kernel is deterministic run to run, loss included
    idx_2, zq_2, loss_2 = native.pq_assign(code.to(dev), books.to(dev))
    assert torch.equal(idx_2, idx_s) and torch.equal(zq_2, zq_s) and torch.equal(loss_2, loss_s)
    if k in (256, 512):   # these sizes have two screened kernels (scores kept in registers / scored twice): same bits
        import ctypes
        try:
            native.load_library().equss_dev_pq_mode(ctypes.c_int(2))
            idx_3, zq_3, loss_3 = native.pq_assign(code.to(dev), books.to(dev))
        finally:
            _force_exact(native, False)
        assert torch.equal(idx_3, idx_s) and torch.equal(zq_3, zq_s)
        assert torch.allclose(loss_3, loss_s, rtol=1e-5)   # (per-workgroup partial sums: other wave counts)


def test_training_quantiser_gradients_match_autograd(pkg, dev):
    """The closed-form backward of the training-pass quantiser against autograd through the stock
    formulation (dino_pqgo.py:613-690) on the same indices: loss, straight-through output, gradients
    w.r.t. the code rows and the codebooks."""
    import torch.nn.functional as F
    torch.manual_seed(2)
    n, num_pq, k, d = 3000, 8, 256, 128
    rows = torch.randn(n, num_pq * d, device=dev).requires_grad_(True)
    books = (torch.rand(num_pq, k, d, device=dev) * 2 - 1).requires_grad_(True)
    beta, book = 0.25, 1.0
    g_out = torch.randn(n, num_pq * d, device=dev)
    zq, vq, idx = pkg.pqgo._PQStraightThrough.apply(rows, books, beta, book)
    got = torch.autograd.grad((zq * g_out).sum() + 3.0 * vq, (rows, books))
    z3 = rows.reshape(n, num_pq, d)
    zn = F.normalize(z3, dim=-1)
    emb = torch.gather(books, 1, idx.unsqueeze(-1).expand(num_pq, n, d)).permute(1, 0, 2)
    cb = ((emb - zn.detach()) ** 2).mean(dim=(0, 2))
    commit = ((zn - emb.detach()) ** 2).mean(dim=(0, 2))
    vq_ref = (book * cb + beta * commit).mean()
    zq_ref = (zn + (emb - zn).detach()).reshape(n, num_pq * d)
    ref = torch.autograd.grad((zq_ref * g_out).sum() + 3.0 * vq_ref, (rows, books))
    assert abs(float(vq.detach()) - float(vq_ref.detach())) < 1e-5 * abs(float(vq_ref.detach()))
    assert float((zq - zq_ref).abs().max()) < 1e-6
    for name, a, b in zip(("rows", "books"), got, ref):
        assert float((a - b).norm() / b.norm()) < 1e-5, name


def test_rows_with_nan_or_inf_get_index_zero(native, dev):
    """Every distance of a row that holds a NaN (or an Inf, which l2 normalisation turns into NaN) is NaN;
    torch.argmin then returns 0, the first NaN, and so do both kernels.  Finite rows beside it are untouched."""
    import ctypes
    rng = np.random.Generator(np.random.PCG64(77))
    code = torch.from_numpy(rng.standard_normal((48, 1024)).astype(np.float32))
    books = torch.from_numpy(rng.uniform(-1 / 256, 1 / 256, size=(8, 256, 128)).astype(np.float32))
    code[5, 130] = float("nan")          # sub-space 1 of row 5
    code[17, 3] = float("inf")           # sub-space 0 of row 17
    z = torch.nn.functional.normalize(code.reshape(48, 8, 128), dim=-1)
    cn = torch.nn.functional.normalize(books, dim=-1)
    dist = (z ** 2).sum(-1, keepdim=True).permute(1, 0, 2) + (cn ** 2).sum(-1).unsqueeze(1) - 2 * torch.einsum("npd,pkd->pnk", z, cn)
    want = dist.argmin(dim=-1)
    assert want[1, 5] == 0 and want[0, 17] == 0
    for mode in (0, 1, 2):
        native.load_library().equss_dev_pq_mode(ctypes.c_int(mode))
        try:
            idx, zq, loss = native.pq_assign(code.to(dev), books.to(dev))
        finally:
            native.load_library().equss_dev_pq_mode(ctypes.c_int(0))
        got = idx.cpu()
        assert got[1, 5] == 0 and got[0, 17] == 0
        mask = torch.ones(8, 48, dtype=torch.bool)
        mask[1, 5] = mask[0, 17] = False
        assert torch.equal(got[mask], want[mask])
