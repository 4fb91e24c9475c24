"""Each HIP building block of the ViT / head forward against a plain PyTorch fp32 reference of the
same op on CPU (operands rounded to bf16 first where the kernel consumes bf16, so the comparison
isolates the kernel's own arithmetic)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(300, 256, 192), (128, 128, 64), (1, 384, 384), (1000, 1152, 384)])
def test_gemm_exact_on_integer_data(epi, shape, native, dev):
    """Small-integer operands make every product and partial sum exact in bf16/fp32, so any fragment
    or layout mistake (swapped rows/cols, wrong k order, missed tail) shows up as a hard mismatch."""
    m, n, k = shape
    g = _rng(epi * 100 + m)
    a = torch.from_numpy(g.integers(-3, 4, size=(m, k)).astype(np.float32))
    w = torch.from_numpy(g.integers(-2, 3, size=(n, k)).astype(np.float32))
    w[:, 0] += torch.arange(n, dtype=torch.float32) % 5          # asymmetric in n
    a[:, 1] += (torch.arange(m, dtype=torch.float32) % 3)        # asymmetric in m
    bias = torch.from_numpy(g.integers(-4, 5, size=(n,)).astype(np.float32))
    ref = a @ w.t() + bias
    if epi == 1:
        ref = F.gelu(ref)
    elif epi == 2:
        ref = F.relu(ref)
    out = None
    if epi == 4:
        base = torch.from_numpy(g.integers(-8, 9, size=(m, n)).astype(np.float32))
        ref = base + ref
        out = base.clone().to(dev)
    got = native.gemm_bf16(_bf(a).to(dev), _bf(w).to(dev), bias.to(dev), epi, out=out)
    torch.cuda.synchronize()
    if epi in (3, 4):
        assert got.dtype == torch.float32
        assert torch.equal(got.cpu(), ref)
    else:
        assert got.dtype == torch.bfloat16
        torch.testing.assert_close(got.cpu().float(), _bf(ref).float(), rtol=8e-3, atol=1e-6)


@pytest.mark.parametrize("n", [128, 256])
def test_gelu_epilogue_accuracy(n, native, dev):
    """The epilogue's GELU against torch's erf form in fp64 over the whole useful range: the
    pre-activation is x = hi + lo (two bf16 columns against unit weights, exact in fp32), the bf16
    result is the correctly rounded one up to 1e-6 absolute, and equal to it for nearly all x."""
    x = torch.cat([torch.linspace(-9.0, 9.0, 8190), torch.tensor([0.0, 1e-3, -1e-3, 20.0, -20.0, 100.0, -100.0,
                                                                 1e-8, -1e-8, 3e4])])
    hi = _bf(x)
    lo = _bf(x - hi.float())
    xe = hi.double() + lo.double()
    a = torch.zeros(x.numel(), 64, dtype=torch.bfloat16)
    a[:, 0], a[:, 1] = hi, lo
    w = torch.zeros(n, 64, dtype=torch.bfloat16)
    w[:, :2] = 1.0
    got = native.gemm_bf16(a.to(dev), w.to(dev), None, 1).cpu()
    assert torch.equal(got[:, :1].expand(-1, n), got)            # every column saw the same x
    g = got[:, 0].double()
    ref = xe * 0.5 * torch.erfc(-xe / math.sqrt(2.0))
    ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(2.0 ** -126))) - 7)   # bf16 spacing at ref
    err = (g - ref).abs()
    worst = (err - 0.5 * ulp).max()
    assert float(worst) <= 1e-6, float(worst)                      # correct rounding + 1e-6 for the formula
    exact = _bf(ref.float()).double()
    big = ref.abs() > 1e-2                                         # (the far negative tail is only absolutely accurate)
    assert float((g[big] == exact[big]).double().mean()) > 0.99


def test_gemm_random_and_strided(native, dev):
    g = _rng(5)
    m, n, k = 777, 384, 1536
    a = _bf(torch.from_numpy(g.standard_normal((m, k + 64)).astype(np.float32)))
    w = _bf(torch.from_numpy((g.standard_normal((n, k)) / math.sqrt(k)).astype(np.float32)))
    bias = torch.from_numpy(g.standard_normal(n).astype(np.float32))
    a_dev = a.to(dev)
    got = native.gemm_bf16(a_dev[:, :k], w.to(dev), bias.to(dev), 3)
    ref = a[:, :k].double() @ w.double().t() + bias.double()
    assert (got.cpu().double() - ref).abs().max() < 2e-4
    got_nobias = native.gemm_bf16(a_dev[:, :k], w.to(dev), None, 3)
    assert (got_nobias.cpu().double() - (ref - bias.double())).abs().max() < 2e-4


def test_gemm_rejects_bad_shapes(native, dev):
    a = torch.zeros(8, 64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        native.gemm_bf16(a, torch.zeros(100, 64, dtype=torch.bfloat16, device=dev), None, 3)   # N % 128
    with pytest.raises(ValueError):
        native.gemm_bf16(a[:, :32], torch.zeros(128, 32, dtype=torch.bfloat16, device=dev), None, 3)  # K % 64


@pytest.mark.parametrize("c", [384, 768])
def test_layernorm(c, native, dev):
    g = _rng(c)
    x = torch.from_numpy((3 * g.standard_normal((10, c)) + 1.5).astype(np.float32))
    gamma = torch.from_numpy((1 + 0.1 * g.standard_normal(c)).astype(np.float32))
    beta = torch.from_numpy((0.1 * g.standard_normal(c)).astype(np.float32))
    ref = F.layer_norm(x, (c,), gamma, beta, 1e-6)
    out, out32 = native.layernorm_bf16(x.to(dev), gamma.to(dev), beta.to(dev), 1e-6, want_f32=True)
    torch.testing.assert_close(out32.cpu(), ref, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(out.cpu().float(), _bf(ref).float(), rtol=8e-3, atol=1e-6)
    # CLS drop: 2 images x 5 tokens -> 8 rows
    out = native.layernorm_bf16(x.to(dev), gamma.to(dev), beta.to(dev), 1e-6, tokens_per_image=5)
    keep = [1, 2, 3, 4, 6, 7, 8, 9]
    assert out.shape == (8, c)
    torch.testing.assert_close(out.cpu().float(), _bf(ref[keep]).float(), rtol=8e-3, atol=1e-6)


def _attention_ref(qkv, b, n, h):
    q, k, v = qkv.double().reshape(b, n, 3, h, 64).permute(2, 0, 3, 1, 4)
    att = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
    return (att @ v).transpose(1, 2).reshape(b * n, h * 64)


@pytest.mark.parametrize("b,n,h", [(2, 65, 6), (1, 200, 2), (1, 1601, 1), (3, 64, 12), (1, 129, 6)])
def test_attention_matches_fp64_reference(b, n, h, native, dev):
    g = _rng(n + h)
    qkv = _bf(torch.from_numpy(g.standard_normal((b * n, 3 * h * 64)).astype(np.float32)))
    ref = _attention_ref(qkv, b, n, h)
    out = native.attention_bf16(qkv.to(dev), b, n, h, 0.125)
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs()
    assert err.max() < 2e-2, float(err.max())
    assert err.mean() < 2e-3, float(err.mean())


def test_attention_exact_layout_on_selector_data(native, dev):
    """One-hot attention: every query matches exactly one key by a huge margin, so the output row
    must equal that key's V row (bf16-exact).  Catches any key/query/d permutation mistake."""
    b, n, h = 1, 150, 2
    g = _rng(9)
    perm = torch.from_numpy(g.permutation(n))
    qkv = torch.zeros(n, 3, h, 64)
    code = torch.from_numpy(g.choice([-1.0, 1.0], size=(n, 64)).astype(np.float32))  # near-orthogonal +-1 codes
    for hh in range(h):
        qkv[:, 1, hh] = code * 4                   # keys
        qkv[:, 0, hh] = code[perm] * 4             # query i looks for key perm[i]
        qkv[:, 2, hh] = torch.from_numpy(g.integers(-8, 9, size=(n, 64)).astype(np.float32))
    out = native.attention_bf16(_bf(qkv.reshape(n, -1)).to(dev), b, n, h, 0.125).cpu().float().reshape(n, h, 64)
    want = qkv[:, 2][perm]                          # (n, h, 64)
    assert (out - want).abs().max() < 0.05


def test_attention_rescale_branch_is_exercised(native, dev):
    """A key far above the running max in a late tile forces the online-softmax rescale."""
    b, n, h = 1, 300, 1
    g = _rng(10)
    qkv = torch.from_numpy(g.standard_normal((n, 3, 1, 64)).astype(np.float32)) * 0.5
    qkv[7, 0, 0] = 0.0
    qkv[7, 0, 0, :8] = 6.0
    qkv[250, 1, 0, :8] = 6.0      # tile 3 holds the spike for query 7
    qkv = _bf(qkv.reshape(n, -1))
    ref = _attention_ref(qkv, b, n, h)
    out = native.attention_bf16(qkv.to(dev), b, n, h, 0.125)
    assert (out.cpu().double() - ref).abs().max() < 2e-2


@pytest.mark.parametrize("gain", [6.0, 24.0])
def test_attention_large_logits(gain, native, dev):
    """Logits of tens to thousands (log2 units): the softmax reference is kept in bf16 and rounded up,
    so exp2 never overflows and the result stays a convex combination of V rows."""
    b, n, h = 1, 333, 2
    g = _rng(12)
    qkv = torch.from_numpy(g.standard_normal((n, 3, h, 64)).astype(np.float32))
    qkv[:, :2] *= gain
    qkv[:64, 1] *= 0.01            # the first key tile scores far below the later ones
    qkv = _bf(qkv.reshape(n, -1))
    ref = _attention_ref(qkv, b, n, h)
    out = native.attention_bf16(qkv.to(dev), b, n, h, 0.125).cpu().double()
    assert torch.isfinite(out).all()
    # q is rounded to bf16 once more after the scale is folded in: logits of magnitude L carry ~L * 2^-9
    # of extra noise (L ~ 150 / 2400 here), which shows where two keys compete closely
    err = (out - ref).abs()
    assert err.max() < (0.3 if gain < 10 else 1.5) and err.mean() < (5e-3 if gain < 10 else 2e-2), (float(err.max()), float(err.mean()))
    assert out.abs().max() <= qkv.reshape(n, 3, h, 64)[:, 2].abs().max().double() + 1e-2


def test_patch_tokens(native, dev):
    g = _rng(11)
    b, hh, ww, p, c = 2, 48, 80, 8, 384
    img = torch.from_numpy(g.standard_normal((b, 3, hh, ww)).astype(np.float32))
    w = torch.from_numpy((0.07 * g.standard_normal((c, 3, p, p))).astype(np.float32))
    bias = torch.from_numpy((0.05 * g.standard_normal(c)).astype(np.float32))
    cls = torch.from_numpy(g.standard_normal(c).astype(np.float32))
    n = (hh // p) * (ww // p)
    pos = torch.from_numpy(g.standard_normal((n + 1, c)).astype(np.float32))
    ref = F.conv2d(_bf(img).float(), _bf(w).float(), bias, stride=p).flatten(2).transpose(1, 2)
    ref = torch.cat((cls.expand(b, 1, c), ref), dim=1) + pos
    x = native.patch_tokens(img.to(dev), _bf(w.reshape(c, -1)).to(dev), bias.to(dev), cls.to(dev), pos.to(dev), p)
    torch.testing.assert_close(x.cpu().reshape(b, n + 1, c), ref, rtol=1e-4, atol=1e-4)


# the last case is one full round of 256 row tiles plus a short remainder, which the library routes
# through a second, smaller-tile launch
@pytest.mark.parametrize("m,k", [(300, 384), (128, 1536), (1000, 384), (256 * 128 + 200, 384)])
def test_gemm_residual_layernorm_fused(m, k, native, dev):
    """x += a w^T + b; ln = LayerNorm(x): against fp32 torch on bf16-rounded operands."""
    g = _rng(m + k)
    n = 384
    a = _bf(torch.from_numpy(g.standard_normal((m, k)).astype(np.float32)))
    w = _bf(torch.from_numpy((g.standard_normal((n, k)) / math.sqrt(k)).astype(np.float32)))
    bias = torch.from_numpy(g.standard_normal(n).astype(np.float32))
    x0 = torch.from_numpy((2 * g.standard_normal((m, n)) + 0.5).astype(np.float32))
    gamma = torch.from_numpy((1 + 0.1 * g.standard_normal(n)).astype(np.float32))
    beta = torch.from_numpy((0.1 * g.standard_normal(n)).astype(np.float32))
    x_ref = x0.double() + a.double() @ w.double().t() + bias.double()
    ln_ref = F.layer_norm(x_ref, (n,), gamma.double(), beta.double(), 1e-6)
    x = x0.clone().to(dev)
    ln = native.gemm_resid_ln_bf16(a.to(dev), w.to(dev), bias.to(dev), x, gamma.to(dev), beta.to(dev), 1e-6)
    torch.cuda.synchronize()
    assert (x.cpu().double() - x_ref).abs().max() < 1e-4
    torch.testing.assert_close(ln.cpu().float(), ln_ref.float(), rtol=1e-2, atol=1e-2)
    assert (ln.cpu().double() - ln_ref).abs().mean() < 2e-3
    x2 = x0.clone().to(dev)
    assert native.gemm_resid_ln_bf16(a.to(dev), w.to(dev), bias.to(dev), x2, want_ln=False) is None
    assert torch.equal(x2, x)
    with pytest.raises(ValueError):
        native.gemm_resid_ln_bf16(a.to(dev), _bf(torch.zeros(256, k)).to(dev), None, torch.zeros(m, 256, device=dev), want_ln=False)


@pytest.mark.parametrize("m", [1, 130, 128 * 5 + 77, 128 * 300 + 5])
def test_mlp_fused_equals_the_two_gemm_route(m, native, dev):
    """fc1 + GELU + fc2 + residual + LayerNorm in one kernel against the two-launch route
    (equss_gemm_bf16 with the GELU epilogue, then equss_gemm_resid_ln_bf16): same operand rounding and
    the same fp32 operation order per element, so the results are compared bit for bit; plus an fp64
    reference of the whole expression."""
    g = _rng(m)
    c, hid = 384, 1536
    a = _bf(torch.from_numpy(g.standard_normal((m, c)).astype(np.float32)))
    w1 = _bf(torch.from_numpy((g.standard_normal((hid, c)) / math.sqrt(c)).astype(np.float32)))
    b1 = torch.from_numpy((0.3 * g.standard_normal(hid)).astype(np.float32))
    w2 = _bf(torch.from_numpy((g.standard_normal((c, hid)) / math.sqrt(hid)).astype(np.float32)))
    b2 = torch.from_numpy((0.3 * g.standard_normal(c)).astype(np.float32))
    x0 = torch.from_numpy(g.standard_normal((m, c)).astype(np.float32))
    gamma = torch.from_numpy((1 + 0.1 * g.standard_normal(c)).astype(np.float32))
    beta = torch.from_numpy((0.1 * g.standard_normal(c)).astype(np.float32))
    d = lambda t: t.to(dev)   # noqa: E731
    x_f = d(x0.clone())
    ln_f = native.mlp_resid_ln_bf16(d(a), d(w1), d(b1), d(w2), d(b2), x_f, d(gamma), d(beta), 1e-6)
    x_u = d(x0.clone())
    hidden = native.gemm_bf16(d(a), d(w1), d(b1), 1)
    ln_u = native.gemm_resid_ln_bf16(hidden, d(w2), d(b2), x_u, d(gamma), d(beta), 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(x_f, x_u)
    assert torch.equal(ln_f, ln_u)
    h64 = F.gelu(a.double() @ w1.double().t() + b1.double())
    ref = x0.double() + _bf(h64.float()).double() @ w2.double().t() + b2.double()
    assert float((x_f.cpu().double() - ref).abs().max()) < 2e-2     # (bf16 rounding of the hidden activation flips at ties)
    assert float((x_f.cpu().double() - ref).abs().mean()) < 2e-4
    x_n = d(x0.clone())
    assert native.mlp_resid_ln_bf16(d(a), d(w1), d(b1), d(w2), d(b2), x_n, want_ln=False) is None
    assert torch.equal(x_n, x_f)


@pytest.mark.parametrize("m", [1, 128 * 3 + 77, 128 * 300 + 5])
def test_block_tail_fused_equals_separate_launches(m, native, dev):
    """proj + residual + LayerNorm2 + MLP + residual + LayerNorm in one kernel against
    equss_gemm_resid_ln_bf16 followed by equss_mlp_resid_ln_bf16: bit for bit."""
    g = _rng(1000 + m)
    c, hid = 384, 1536
    rnd = lambda *shape, s=1.0: torch.from_numpy((s * g.standard_normal(shape)).astype(np.float32))   # noqa: E731
    att = _bf(rnd(m, c))
    wp, bp = _bf(rnd(c, c, s=1 / math.sqrt(c))), rnd(c, s=0.3)
    w1, b1 = _bf(rnd(hid, c, s=1 / math.sqrt(c))), rnd(hid, s=0.3)
    w2, b2 = _bf(rnd(c, hid, s=1 / math.sqrt(hid))), rnd(c, s=0.3)
    x0 = rnd(m, c)
    g2, be2 = 1 + rnd(c, s=0.1), rnd(c, s=0.1)
    g1, be1 = 1 + rnd(c, s=0.1), rnd(c, s=0.1)
    d = lambda t: t.to(dev)   # noqa: E731
    x_f = d(x0.clone())
    mid_f, ln_f = native.block_tail_bf16(d(att), d(wp), d(bp), d(g2), d(be2), d(w1), d(b1), d(w2), d(b2), x_f, d(g1), d(be1))
    x_u = d(x0.clone())
    mid_u = native.gemm_resid_ln_bf16(d(att), d(wp), d(bp), x_u, d(g2), d(be2), 1e-6)
    ln_u = native.mlp_resid_ln_bf16(mid_u, d(w1), d(b1), d(w2), d(b2), x_u, d(g1), d(be1), 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(mid_f, mid_u)
    assert torch.equal(x_f, x_u)
    assert torch.equal(ln_f, ln_u)
    x_n = d(x0.clone())
    _, none = native.block_tail_bf16(d(att), d(wp), d(bp), d(g2), d(be2), d(w1), d(b1), d(w2), d(b2), x_n, want_ln=False)
    assert none is None and torch.equal(x_n, x_f)


@pytest.mark.parametrize("shape", [(20000, 768, 768), (17000, 768, 3072), (40000, 384, 1536)])
def test_residual_gemm_large_tile_route_equals_small_tile_route(shape, native, dev):
    """out += a w^T + bias: the large-tile kernel (long M) and the 128 x 128 kernel (short M, trailing
    rows) apply the same fp32 operations in the same order, so a row's bits do not depend on the route."""
    import ctypes
    m, n, k = shape
    g = _rng(m + n)
    a = _bf(torch.from_numpy(g.standard_normal((m, k)).astype(np.float32))).to(dev)
    w = _bf(torch.from_numpy((g.standard_normal((n, k)) / math.sqrt(k)).astype(np.float32))).to(dev)
    bias = torch.from_numpy(g.standard_normal(n).astype(np.float32)).to(dev)
    x0 = torch.from_numpy(g.standard_normal((m, n)).astype(np.float32)).to(dev)
    lib = native.load_library()
    big = native.gemm_bf16(a, w, bias, 4, out=x0.clone())
    lib.equss_dev_gemm_variant(ctypes.c_int(0))
    try:
        small = native.gemm_bf16(a, w, bias, 4, out=x0.clone())
    finally:
        lib.equss_dev_gemm_variant(ctypes.c_int(3))
    torch.cuda.synchronize()
    assert torch.equal(big, small)
    ref = x0.double() + a.double() @ w.double().t() + bias.double()
    assert float((big.double() - ref).abs().max()) < 5e-4


@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("shape", [(16384 + 57, 1152, 384), (20000, 384, 768), (16384, 768, 64)])
def test_producer_consumer_gemm_equals_the_one_role_kernels(epi, shape, native, dev):
    """Long M, N a multiple of 384, bf16 output: the producer / consumer kernel (four request-only waves feed
    eight MFMA waves) applies the same fp32 operations in the same order as the large-tile and the 128 x 128
    kernels: equal bits, ragged last row tile and single-K-step-pair shapes included."""
    import ctypes
    m, n, k = shape
    g = _rng(7 * m + n + epi)
    a = _bf(torch.from_numpy(g.standard_normal((m, k)).astype(np.float32))).to(dev)
    w = _bf(torch.from_numpy((g.standard_normal((n, k)) / math.sqrt(k)).astype(np.float32))).to(dev)
    bias = torch.from_numpy(g.standard_normal(n).astype(np.float32)).to(dev)
    lib = native.load_library()
    guard = torch.full((m + 8, n), 7.0, dtype=torch.bfloat16, device=dev)      # rows past M must stay untouched
    ws = native.gemm_bf16(a, w, bias, epi, out=guard[:m])
    outs = []
    for variant in (6, 4, 0):   # (6: the default route without the stationary-weights kernel, which takes K = 384)
        lib.equss_dev_gemm_variant(ctypes.c_int(variant))
        try:
            outs.append(native.gemm_bf16(a, w, bias, epi))
        finally:
            lib.equss_dev_gemm_variant(ctypes.c_int(3))
    torch.cuda.synchronize()
    assert torch.equal(ws, outs[0]) and torch.equal(ws, outs[1]) and torch.equal(ws, outs[2])
    assert bool((guard[m:] == 7.0).all())
    ref = a.double() @ w.double().t() + bias.double()
    ref = F.gelu(ref) if epi == 1 else F.relu(ref) if epi == 2 else ref
    assert float((ws.double() - ref).abs().max()) < 0.05


@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("shape", [(16384 + 57, 1152), (102464, 1152), (20000 + 5, 384), (16384 + 1, 768), (51200, 1536)])
def test_stationary_weights_gemm_equals_the_ring_kernels(epi, shape, native, dev):
    """K = 384, long M, N a multiple of 384 (qkv, the head's first GEMM): the kernel that keeps a 384-column strip of W
    in registers for the whole launch gives the bits of the producer / consumer and large-tile kernels; ragged last
    row block, row counts that leave some workgroups without work, one to four strips, strided output rows."""
    import ctypes
    m, n = shape
    k = 384
    g = _rng(11 * m + n + epi)
    a = _bf(torch.from_numpy(g.standard_normal((m, k)).astype(np.float32))).to(dev)
    w = _bf(torch.from_numpy((g.standard_normal((n, k)) / math.sqrt(k)).astype(np.float32))).to(dev)
    bias = torch.from_numpy(g.standard_normal(n).astype(np.float32)).to(dev)
    lib = native.load_library()
    guard = torch.full((m + 8, n + 8), 7.0, dtype=torch.bfloat16, device=dev)   # rows past M and the padding columns stay untouched
    sw = native.gemm_bf16(a, w, bias, epi, out=guard[:m, :n])
    lib.equss_dev_gemm_variant(ctypes.c_int(6))
    try:
        ring = native.gemm_bf16(a, w, bias, epi)
    finally:
        lib.equss_dev_gemm_variant(ctypes.c_int(3))
    torch.cuda.synchronize()
    assert torch.equal(sw, ring)
    assert bool((guard[m:] == 7.0).all()) and bool((guard[:, n:] == 7.0).all())
    ref = a.double() @ w.double().t() + bias.double()
    ref = F.gelu(ref) if epi == 1 else F.relu(ref) if epi == 2 else ref
    assert float((sw.double() - ref).abs().max()) < 0.05


def test_gemm_rejects_a_bf16_output_stride_that_breaks_16_byte_rows(native, dev):
    """bf16 output rows go out in 16-byte pieces: a leading dimension that is a multiple of 4 but not of 8 is refused
    (it is fine for an fp32 output)."""
    a = torch.zeros((256, 64), dtype=torch.bfloat16, device=dev)
    w = torch.zeros((128, 64), dtype=torch.bfloat16, device=dev)
    out = torch.zeros((256, 132), dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError, match="multiple of 8"):
        native.gemm_bf16(a, w, None, 0, out=out[:, :128])
    out32 = torch.zeros((256, 132), dtype=torch.float32, device=dev)
    native.gemm_bf16(a, w, None, 3, out=out32[:, :128])
    torch.cuda.synchronize()
    assert float(out32.abs().max()) == 0.0
