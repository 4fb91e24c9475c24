"""Properties at the benchmark's full size (64 images of 320x320: 102 400 rows x 1024 channels, 8 x 256
codewords), where no CPU oracle finishes in seconds: idempotence, independence of rows and of images
(what sharding images over ranks relies on), permutation equivariance, conservation of counts."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_case(dev):
    gen = torch.Generator(device="cpu").manual_seed(5)
    code = torch.randn(64 * 1600, 1024, generator=gen).to(dev)
    books = (torch.rand(8, 256, 128, generator=gen) - 0.5).to(dev)
    return code, books


def test_pq_full_size_properties(big_case, native, dev):
    code, books = big_case
    rows = code.shape[0]
    idx, zq, loss = native.pq_assign(code, books)
    assert idx.shape == (8, rows) and int(idx.min()) >= 0 and int(idx.max()) < 256
    # z_q is the chosen raw codeword (straight-through form), sub-space by sub-space
    gathered = torch.stack([books[p][idx[p]] for p in range(8)], dim=1).reshape(rows, 1024)
    assert (zq - gathered).abs().max() < 3e-7
    # conservation: the usage histogram of every sub-space sums to the number of rows
    for p in range(8):
        assert int(torch.bincount(idx[p], minlength=256).sum()) == rows
    # idempotence: a codeword's nearest codeword is itself
    idx2, zq2, _ = native.pq_assign(gathered, books)
    assert torch.equal(idx2, idx)
    # rows are independent: two halves, and a permutation of the rows, give the same assignment
    a, _, la = native.pq_assign(code[: rows // 2 + 7], books)
    b, _, lb = native.pq_assign(code[rows // 2 + 7:], books)
    assert torch.equal(torch.cat([a, b], dim=1), idx)
    perm = torch.randperm(rows, generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)
    ip, zp, lp = native.pq_assign(code[perm], books)
    assert torch.equal(ip, idx[:, perm]) and torch.equal(zp, zq[perm])
    # loss = sum of squared (codeword - normalised row); split sums add up, permutation changes only rounding
    zn = torch.nn.functional.normalize(code.view(rows, 8, 128), dim=2)
    want = ((gathered.view(rows, 8, 128) - zn) ** 2).sum(dim=(0, 2))
    assert torch.allclose(loss, want, rtol=1e-4)
    assert torch.allclose(la + lb, loss, rtol=1e-5) and torch.allclose(lp, loss, rtol=1e-5)
    # bitwise reproducible run to run
    i3, z3, l3 = native.pq_assign(code, books)
    assert torch.equal(i3, idx) and torch.equal(z3, zq) and torch.equal(l3, loss)


def test_model_full_batch_equals_per_shard_results(pkg, dev):
    """One 64-image batch against the same images run as shards of 16 (what data-parallel ranks do).
    Images are independent and every kernel applies the same fp32 operations in the same order to a
    row whatever the batch -- including the trailing rows of the residual GEMMs, which take a
    small-tile route with a LayerNorm launch of its own -- so results are bit-identical image by image."""
    cfg = pkg.configs.S_CFG
    m = pkg.build_model(cfg)
    m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    m = m.to(dev).eval()
    img = pkg.synth.synth_images(64, 320, 320, seed=3).to(dev)
    with torch.no_grad():
        code, quant, (_s, _q, index), out = m.model(img=img)
        code, quant, index = code.clone(), quant.clone(), index.clone()
        for s in range(0, 64, 16):
            c2, q2, (_a, _b, i2), _o = m.model(img=img[s:s + 16])
            assert torch.equal(i2, index[:, s:s + 16]), f"indices differ for shard {s}"
            assert torch.equal(c2, code[s:s + 16]) and torch.equal(q2, quant[s:s + 16])
    assert index.shape == (8, 64, 40, 40)
    assert torch.isfinite(code).all()


def test_streamed_eval_forward_equals_single_stream(pkg, dev):
    """The eval forward spread over two or three HIP streams (image ranges of the batch) gives the
    same code, quantised map and indices, bit for bit, as the single-stream forward."""
    cfg = pkg.configs.S_CFG
    m = pkg.build_model(cfg)
    m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    m = m.to(dev).eval()
    img = pkg.synth.synth_images(7, 96, 64, seed=21).to(dev)
    with torch.no_grad():
        m.model.eval_streams = 1
        m.model.eval_stream_min_images = 1
        ref = m.model(img=img)
        for k in (2, 3):
            m.model.eval_streams = k
            for _ in range(2):   # second call: cached streams and workspaces
                got = m.model(img=img)
            torch.cuda.synchronize()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2][2], ref[2][2]), k
            assert all(torch.equal(a, b) for a, b in zip(got[2][1], ref[2][1]))
            assert abs(float(got[3]["vq-loss"]) - float(ref[3]["vq-loss"])) < 1e-6 * abs(float(ref[3]["vq-loss"])) + 1e-9


def test_bcfg_at_its_real_shape_768px(pkg, native, oracle, dev):
    """ViT-B/8 + 16 x 512 PQ on 2 crops of 768 x 768 (N = 9 217 tokens per image, the shape the configuration
    is defined by): (a) one attention head at that sequence length against an fp64 softmax(q k^T) v,
    (b) the PQ indices of the path's own 18 432 code rows against the fp64 arg-min outside near-ties,
    (c) a batch of two gives each image the bits it gets alone."""
    import numpy as np
    from conftest import assert_index_parity
    cfg = pkg.configs.B_CFG
    m = pkg.build_model(cfg)
    state = pkg.synth.synth_pqgo_state(cfg, seed=1234)
    m.load_state_dict(state, strict=True)
    m = m.to(dev).eval()
    m.model.eval_streams = 1
    img = pkg.synth.synth_images(2, 768, 768, seed=31).to(dev)
    with torch.no_grad():
        code, quant, (_s, _q, index), out = m.model(img=img)
        code, quant, index = code.clone(), quant.clone(), index.clone()
        for i in range(2):
            c1, q1, (_a, _b, i1), _o = m.model(img=img[i:i + 1])
            assert torch.equal(i1, index[:, i:i + 1]) and torch.equal(c1, code[i:i + 1]) and torch.equal(q1, quant[i:i + 1])
    assert index.shape == (16, 2, 96, 96) and torch.isfinite(code).all()
    # (b) indices of the path's own code rows
    rows = code.permute(0, 2, 3, 1).reshape(-1, 2048).cpu().contiguous()
    books = [state[f"model.vq_blocks.0.quantizers.{i}.embedding.weight"] for i in range(16)]
    want, gap = oracle.pq_top2_gap(books, rows, "l2")
    flips = assert_index_parity(index.reshape(16, -1).cpu().numpy(), want.numpy(), gap.numpy(), "B-cfg 768px own code")
    print("B-cfg 2 x 768^2: near-tie flips on own code:", flips, "of", want.numel())
    # (a) attention at N = 9217: 12 heads computed, head 5 of image 1 checked in fp64
    n, heads = 9217, 12
    gen = torch.Generator(device="cpu").manual_seed(9)
    qkv = (torch.randn(2 * n, 3 * heads * 64, generator=gen) * 1.5).to(torch.bfloat16).to(dev)
    got = native.attention_bf16(qkv, 2, n, heads, 0.125)
    h = 5
    blk = qkv[n:2 * n].double().reshape(n, 3, heads, 64)
    q, k, v = blk[:, 0, h], blk[:, 1, h], blk[:, 2, h]
    ref = torch.softmax((q @ k.t()) * 0.125, dim=-1) @ v
    err = float((got[n:2 * n, h * 64:(h + 1) * 64].double() - ref).norm() / ref.norm())
    print("attention at N = 9217, rel-L2 vs fp64:", err)
    assert err < 1e-2


def test_block_tail_kernels_give_equal_bits_inside_the_vit(pkg, native, dev):
    """The register-streamed block-tail kernel (the default inside the ViT forward) against the producer / consumer kernel
    (64-row tiles) and the eight-wave kernel + small-tile launches: equal bits of the features, at a row count that leaves a
    ragged last tile (37 x 1601 rows), at one that does not divide into rounds either way, and at a small one where a
    workgroup's tile has fewer than four 16-row groups."""
    import ctypes
    lib = native.load_library()
    torch.manual_seed(3)
    net = pkg.featurizer.vit_small(patch_size=8).to(dev).eval()
    for batch in (37, 48, 5):
        img = torch.randn(batch, 3, 320, 320, device=dev)
        feats = []
        try:
            lib.equss_dev_fuse_min_tiles(ctypes.c_int(1))   # (the fused kernels at every row count)
            for on in (3, 1, 0, 2):   # default, the producer / consumer kernel (eight consumer waves), the eight-wave kernel, four consumer waves
                lib.equss_dev_block_ws(ctypes.c_int(on))
                with torch.no_grad():
                    feats.append(net.get_intermediate_feat(img, n=1)[0][0].clone())
        finally:
            lib.equss_dev_block_ws(ctypes.c_int(3))
            lib.equss_dev_fuse_min_tiles(ctypes.c_int(384))
        assert torch.isfinite(feats[0]).all()
        assert all(torch.equal(feats[0], f) for f in feats[1:])


def test_stationary_weights_gemm_gives_equal_bits_inside_the_vit(pkg, native, dev):
    """qkv of every block (and, at these row counts, nothing else) goes through the stationary-weights GEMM by default:
    the features equal those of the producer / consumer route bit for bit, at a row count with a ragged last row block."""
    import ctypes
    lib = native.load_library()
    torch.manual_seed(5)
    net = pkg.featurizer.vit_small(patch_size=8).to(dev).eval()
    img = torch.randn(37, 3, 320, 320, device=dev)
    feats = []
    try:
        for variant in (3, 6):
            lib.equss_dev_gemm_variant(ctypes.c_int(variant))
            with torch.no_grad():
                feats.append(net.get_intermediate_feat(img, n=1)[0][0].clone())
    finally:
        lib.equss_dev_gemm_variant(ctypes.c_int(3))
    assert torch.isfinite(feats[0]).all()
    assert torch.equal(feats[0], feats[1])
