"""End-to-end parity of the HIP path (ViT -> head -> PQ assign behind the reference's module API)
with the reference outputs frozen in tests/golden.

Staged, because a bf16 ViT cannot reproduce fp32 features bit for bit:
  (i)   PQ assign on IDENTICAL fp32 code -> indices exact (tests/test_pq_gpu.py and below);
  (ii)  features / code of the bf16 path within a stated tolerance of the fp32 reference;
  (iii) end-to-end code-index agreement reported as a rate (asserted above a floor)."""
import numpy as np
import pytest
import torch

from conftest import assert_index_parity

pytestmark = pytest.mark.gpu

# bf16 operands (8 significant bits) through 12 residual blocks: measured relative L2 error of the
# final features is ~5e-3; the bound leaves 4x headroom.
FEAT_REL_L2 = 2e-2
CODE_REL_L2 = 2e-2
# ViT-B/8, 16 x 512 codewords, bf16 end to end against the fp32 reference: measured 0.9915 (13 of 1536 indices); a
# flip needs two codewords closer than the bf16 error of the code.  The floor allows about twice as many.
BCFG_INDEX_FLOOR = 0.98


def _rel_l2(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want) / np.linalg.norm(want))


@pytest.fixture(scope="module")
def wrapper(pkg, dev):
    cfg = pkg.configs.S_CFG
    m = pkg.build_model(cfg)
    m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    return m.to(dev).eval()


def test_vit_features_64px(wrapper, pkg, golden, dev):
    g = golden("g1_vit_64")
    img = pkg.synth.synth_images(2, 64, 64, seed=11).to(dev)
    bb = wrapper.model.extractor.backbone
    bf, f32 = bb.patch_features(img, want_f32=True)
    want = g["feat"][:, 1:, :].reshape(-1, 384)
    err = _rel_l2(f32.cpu().numpy(), want)
    print("G1 feature rel-L2:", err, "max abs:", float(np.abs(f32.cpu().numpy() - want).max()))
    assert err < FEAT_REL_L2
    assert _rel_l2(bf.float().cpu().numpy(), want) < FEAT_REL_L2
    # get_intermediate_feat (vision_transformer.py:227-239): CLS row included, n > 1 keeps norm(x) per block
    feats, attns, qkvs = bb.get_intermediate_feat(img, n=1)
    assert len(feats) == 1 and feats[0].shape == (2, 65, 384) and attns == [] and qkvs == []
    assert _rel_l2(feats[0].cpu().numpy(), g["feat"]) < FEAT_REL_L2
    assert _rel_l2(feats[0][:, 0].cpu().numpy(), g["feat"][:, 0]) < FEAT_REL_L2          # the CLS feature itself
    assert torch.equal(feats[0][:, 1:].reshape(-1, 384), f32)
    feats3, _, _ = bb.get_intermediate_feat(img, n=3)
    assert len(feats3) == 3 and torch.equal(feats3[2], feats[0])
    with pytest.raises(ValueError):
        bb.get_intermediate_feat(img, n=13)


def test_vit_residual_stream_taps_64px(wrapper, pkg, golden, dev):
    """The fp32 residual stream after blocks 0, 5 and 11 against the reference's (golden taps of G1), and
    norm(x) of the block before last against LayerNorm of the reference's tap."""
    g = golden("g1_vit_64")
    img = pkg.synth.synth_images(2, 64, 64, seed=11).to(dev)
    bb = wrapper.model.extractor.backbone
    taps = bb.residual_taps(img)
    assert taps.shape == (12, 2, 65, 384)
    errs = {i: _rel_l2(taps[i].cpu().numpy(), g[f"block{i}"]) for i in (0, 5, 11)}
    print("residual-stream rel-L2 after blocks 0 / 5 / 11:", errs)
    assert errs[0] < 5e-3 and errs[5] < 1e-2 and errs[11] < FEAT_REL_L2
    feats2, _, _ = bb.get_intermediate_feat(img, n=2)
    want = torch.nn.functional.layer_norm(taps[10], (384,), bb.norm.weight, bb.norm.bias, 1e-6)
    assert float((feats2[0] - want).abs().max()) < 1e-4


def test_featurizer_non_square(wrapper, pkg, golden, dev):
    g = golden("g1b_featurizer_48x80")
    img = pkg.synth.synth_images(1, 48, 80, seed=12).to(dev)
    feat = wrapper.model.extractor(img)
    assert feat.shape == (1, 384, 6, 10) and feat.dtype == torch.float32
    assert _rel_l2(feat.cpu().numpy(), g["feat"]) < FEAT_REL_L2
    with pytest.raises(AssertionError):   # dino_featurizer.py:70
        wrapper.model.extractor(torch.zeros(1, 3, 50, 80, device=dev))


def test_full_forward_scfg_b4_320(wrapper, pkg, oracle, golden, dev):
    g = golden("g5_scfg_b4_320")
    cfg = pkg.configs.S_CFG
    img = pkg.synth.synth_images(4, 320, 320, seed=0).to(dev)
    label = pkg.synth.synth_labels(4, 320, 320, cfg["num_classes"], seed=0).to(dev)
    with torch.no_grad():
        code, quant, (z_split, z_quantized, z_index), outputs = wrapper.model(img=img, aug_img=img, img_pos=None, it=-1)
        total, out, (lin_p, clu_p), feat_vqs = wrapper(img, img, label)
    torch.cuda.synchronize()
    # ---- API surface (model/dino_pqgo.py:152-154, wrapper/PQGOWrapper.py:103) -------------------
    assert code.shape == (4, 1024, 40, 40) and quant.shape == (4, 1024, 40, 40)
    assert z_index.shape == (8, 4, 40, 40) and z_index.dtype == torch.int64
    assert len(z_split) == 8 and z_split[0].shape == (4, 128, 40, 40) and len(z_quantized) == 8
    assert sorted(outputs.keys()) == ["vq-loss"]
    assert sorted(out.keys()) == ["cluster-loss", "linear-loss", "loss", "vq-loss"]
    assert total.shape == (1,) and feat_vqs.shape == (4, 1024, 40, 40)
    assert lin_p.shape == (4, 320, 320) and lin_p.dtype == torch.int64 and clu_p.shape == (4, 320, 320)
    # ---- (ii) tolerance on features / code ---------------------------------------------------------
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])   # noqa: E731
    tok = g["tok"]
    feat = wrapper.model.extractor(img)
    e_feat = _rel_l2(rows(feat)[tok].cpu().numpy(), g["feat_tok"])
    e_code = _rel_l2(rows(code)[tok].cpu().numpy(), g["code_tok"])
    print("G5 feature rel-L2:", e_feat, " code rel-L2:", e_code)
    assert e_feat < FEAT_REL_L2 and e_code < CODE_REL_L2
    # ---- (i) the PQ stage on the HIP path's own fp32 code: exact against the fp64 arg-min ----------
    books = [wrapper.model.vq_blocks[0].quantizers[i].embedding.weight.detach().cpu() for i in range(8)]
    code_rows = rows(code).cpu().contiguous()
    want_idx, gap = oracle.pq_top2_gap(books, code_rows, "l2")
    flips = assert_index_parity(z_index.reshape(8, -1).cpu().numpy(), want_idx.numpy(), gap.numpy(), "own code")
    # ---- (iii) end-to-end agreement with the fp32 reference's indices ------------------------------
    agree = float((z_index.cpu().numpy() == g["index"].astype(np.int64)).mean())
    hist = torch.stack([torch.bincount(z_index[i].reshape(-1), minlength=256) for i in range(8)]).cpu().numpy()
    hist_l1 = float(np.abs(hist - g["hist"]).sum() / g["hist"].sum())
    print(f"G5 end-to-end code-index agreement with the fp32 reference: {agree:.4f}; "
          f"usage-histogram L1 distance {hist_l1:.4f}; near-tie flips on own code: {flips}")
    assert agree > 0.985         # measured 0.992
    # losses
    assert abs(float(out["vq-loss"]) - float(g["vq_loss"])) < 2e-2 * float(g["vq_loss"])
    assert abs(float(out["linear-loss"]) - float(g["linear_loss"])) < 3e-2 * abs(float(g["linear_loss"]))
    assert abs(float(out["cluster-loss"]) - float(g["cluster_loss"])) < 3e-2 * abs(float(g["cluster_loss"]))
    pix = g["pix"]
    lin_agree = float((lin_p.reshape(-1)[pix].cpu().numpy() == g["linear_preds"]).mean())
    clu_agree = float((clu_p.reshape(-1)[pix].cpu().numpy() == g["cluster_preds"]).mean())
    print(f"probe agreement: linear {lin_agree:.4f} cluster {clu_agree:.4f}")
    assert lin_agree > 0.99 and clu_agree > 0.93         # measured 0.996 / 0.947


def test_fused_block_kernels_meet_the_golden_fixture_directly(wrapper, pkg, golden, dev):
    """G5 again with the fused block-tail kernel forced on (it is otherwise used from 384 row tiles up, and
    4 x 1601 rows are 51): the kernel that dominates the benchmark step is compared with the reference's
    outputs directly, not only through its bit-identity with the per-GEMM route."""
    import ctypes
    g = golden("g5_scfg_b4_320")
    lib = pkg.native.load_library()
    img = pkg.synth.synth_images(4, 320, 320, seed=0).to(dev)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])   # noqa: E731
    with torch.no_grad():
        code_a, _q, (_s, _z, idx_a), _o = wrapper.model(img=img)
        lib.equss_dev_fuse_min_tiles(ctypes.c_int(1))
        try:
            pkg.native.profile_enable(256)
            code_b, _q, (_s, _z, idx_b), _o = wrapper.model(img=img)
            feat_b = wrapper.model.extractor(img)
            prof = pkg.native.profile_read(reset=True)
        finally:
            pkg.native.profile_enable(0)
            lib.equss_dev_fuse_min_tiles(ctypes.c_int(384))
    assert prof["block_tail"][1] == 24 and prof["gemm_fc1"][1] == 0       # 12 fused launches per ViT pass
    tok = g["tok"]
    assert _rel_l2(rows(feat_b)[tok].cpu().numpy(), g["feat_tok"]) < FEAT_REL_L2
    assert _rel_l2(rows(code_b)[tok].cpu().numpy(), g["code_tok"]) < CODE_REL_L2
    agree = float((idx_b.cpu().numpy() == g["index"].astype(np.int64)).mean())
    print("fused route: end-to-end code-index agreement with the fp32 reference", agree)
    assert agree > 0.985
    assert torch.equal(code_a, code_b) and torch.equal(idx_a, idx_b)       # both routes give a row the same bits


def test_quantized_output_matches_reference_where_indices_agree(wrapper, pkg, golden, dev):
    g = golden("g5_scfg_b4_320")
    img = pkg.synth.synth_images(4, 320, 320, seed=0).to(dev)
    with torch.no_grad():
        code, quant, (_a, _b, z_index), _o = wrapper.model(img=img)
    tok = g["tok"]
    q = quant.permute(0, 2, 3, 1).reshape(-1, 1024)[tok].cpu().numpy().reshape(len(tok), 8, 128)
    want = g["quant_tok"].reshape(len(tok), 8, 128)
    same = (z_index.reshape(8, -1)[:, tok].cpu().numpy() == g["index"].reshape(8, -1)[:, tok]).T   # (tok, 8)
    # z_q = z_norm + (emb - z_norm) equals the raw codeword up to rounding (dino_pqgo.py:689-690)
    np.testing.assert_allclose(q[same], want[same], rtol=0, atol=3e-7)


def test_legacy_eval_py_call_convention(wrapper, pkg, dev):
    cfg = pkg.configs.S_CFG
    img = pkg.synth.synth_images(1, 64, 64, seed=3).to(dev)
    label = pkg.synth.synth_labels(1, 64, 64, cfg["num_classes"], seed=3).to(dev)
    with torch.no_grad():
        out3 = wrapper(img, label)                      # eval.py:74,:212
        out4 = wrapper(img, img, label)                 # train.py:276
    assert len(out3) == 3 and len(out4) == 4
    assert torch.equal(out3[2][0], out4[2][0]) and torch.equal(out3[2][1], out4[2][1])


def test_training_step_runs_and_gives_gradients(pkg, dev):
    cfg = pkg.configs.pqgo_config(need_initialized="uni")
    m = pkg.build_model(cfg)
    m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    m = m.to(dev).train()
    torch.manual_seed(0)
    img = pkg.synth.synth_images(2, 64, 64, seed=1).to(dev)
    pos = pkg.synth.synth_images(2, 64, 64, seed=2).to(dev)
    label = pkg.synth.synth_labels(2, 64, 64, cfg["num_classes"], seed=1).to(dev)
    before = m.model.vq_blocks[0].quantizers[0].embedding.weight.detach().clone()
    with torch.enable_grad():
        total, out, preds, feat_vqs = m(img, img, label, img_pos=pos, it=0)
    assert not torch.equal(before, m.model.vq_blocks[0].quantizers[0].embedding.weight)  # first-call re-init (:589-609)
    for k in ("vq-loss", "loss", "linear-loss", "cluster-loss", "codebook-usage", "stego-loss", "end_time"):
        assert k in out, k
    total.backward()
    trainable = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    assert len(trainable) == 17 and sum(p.numel() for _, p in trainable) == 1253787
    for n, p in trainable:
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert all(p.grad is None for n, p in m.named_parameters() if not p.requires_grad)


def test_bcfg_vit_base_pq16x512(pkg, oracle, golden, dev):
    """The ViT-B/8 configuration (C = 768, 12 heads, D = 2048, 16 x 512 codewords): every GEMM width,
    the 12-head attention and the K = 512 PQ instance, against the reference fixture."""
    g = golden("g11_bcfg_64x96")
    cfg = pkg.configs.B_CFG
    m = pkg.build_model(cfg)
    m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    m = m.to(dev).eval()
    img = pkg.synth.synth_images(1, 64, 96, seed=13).to(dev)
    with torch.no_grad():
        code, quant, (_s, _q, z_index), outputs = m.model(img=img)
        feat = m.model.extractor(img)
    torch.cuda.synchronize()
    assert code.shape == (1, 2048, 8, 12) and z_index.shape == (16, 1, 8, 12)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])   # noqa: E731
    e_feat = _rel_l2(rows(feat).cpu().numpy(), g["feat_rows"])
    e_code = _rel_l2(rows(code).cpu().numpy(), g["code_rows"])
    print("G11 feature rel-L2:", e_feat, " code rel-L2:", e_code)
    assert e_feat < FEAT_REL_L2 and e_code < CODE_REL_L2
    # PQ on the reference's own fp32 code: exact indices
    books = torch.stack([m.model.vq_blocks[0].quantizers[i].embedding.weight.detach() for i in range(16)])
    idx, zq, loss = pkg.native.pq_assign(torch.from_numpy(g["code_rows"]).to(dev), books)
    assert_index_parity(idx.cpu().numpy(), g["index"].reshape(16, -1), g["gap64"], "G11 reference code")
    # PQ on the HIP path's own code: exact against the fp64 arg-min of that code
    want_idx, gap = oracle.pq_top2_gap([b.cpu() for b in books], rows(code).cpu().contiguous(), "l2")
    assert_index_parity(z_index.reshape(16, -1).cpu().numpy(), want_idx.numpy(), gap.numpy(), "G11 own code")
    assert abs(float(outputs["vq-loss"]) - float(g["vq_loss"])) < 2e-2 * float(g["vq_loss"])
    # end to end (bf16 backbone and head against the fp32 reference): the share of the 16 x 96 indices that agree
    agree = float((z_index.reshape(16, -1).cpu().numpy() == g["index"].reshape(16, -1)).mean())
    print("G11 end-to-end index agreement with the fp32 reference:", agree)
    assert agree >= BCFG_INDEX_FLOOR, agree


def test_training_head_matches_fp32_autograd(pkg, dev):
    """The training pass's expansion head (HIP forward, bf16 GEMM backward) against fp32 autograd
    through three F.linear (dino_pqgo.py:104-112): code and every parameter gradient."""
    import torch.nn.functional as F
    torch.manual_seed(5)
    n, c, d = 1000, 384, 1024
    rows = torch.randn(n, c, device=dev)
    ps = [torch.randn(d, c, 1, 1, device=dev) * 0.05, torch.randn(d, device=dev) * 0.1,
          torch.randn(c, c, 1, 1, device=dev) * 0.05, torch.randn(c, device=dev) * 0.1,
          torch.randn(d, c, 1, 1, device=dev) * 0.05, torch.randn(d, device=dev) * 0.1]
    ps = [p.requires_grad_(True) for p in ps]
    g_out = torch.randn(n, d, device=dev)
    code = pkg.pqgo._HeadFunction.apply(rows, *ps)
    got = torch.autograd.grad((code * g_out).sum(), ps)
    w1, b1, w2a, b2a, w2b, b2b = ps
    ref_code = F.linear(rows, w1.view(d, c), b1) + F.linear(F.relu(F.linear(rows, w2a.view(c, c), b2a)), w2b.view(d, c), b2b)
    ref = torch.autograd.grad((ref_code * g_out).sum(), ps)
    assert float(((code - ref_code).norm() / ref_code.norm()).detach()) < 1e-2
    errs = {name: float((a - b).norm() / b.norm()) for name, a, b in zip(("w1", "b1", "w2a", "b2a", "w2b", "b2b"), got, ref)}
    print("head gradient rel-L2 errors:", errs)
    for name, a, b in zip(errs, got, ref):
        assert a.shape == b.shape, name
    # (w2a, b2a: the ReLU mask is the bf16 forward's; the ~0.1 % of units close enough to zero to flip
    # against the fp32 forward give a few percent in the norm)
    assert max(errs.values()) < 5e-2, errs


@pytest.mark.parametrize("rows,m,n", [(1000, 128, 128), (4099, 384, 256), (102400, 1024, 768), (31, 128, 384)])
def test_transposed_operand_gemm_matches_fp64(rows, m, n, native, dev):
    """equss_gemm_tn_bf16 (C = A^T B summed over the rows, split over workgroups, operands through the transposing
    LDS read) against the fp64 product of the same bf16 operands: row counts that are no multiple of the 32-row K
    step or of the split, a strided B, and the head's weight-gradient shape at the benchmark size."""
    gen = torch.Generator(device="cpu").manual_seed(rows)
    a = torch.randn(rows, m, generator=gen).to(torch.bfloat16).to(dev)
    b_full = torch.randn(rows, n + 64, generator=gen).to(torch.bfloat16).to(dev)
    b = b_full[:, 64:] if rows < 50000 else b_full[:, :n]
    got = native.gemm_tn_bf16(a, b)
    if rows > 50000:   # fp64 of the whole product is slow: a slice of the output
        want = (a[:, :256].double().t() @ b[:, :128].double()).float()
        got = got[:256, :128]
    else:
        want = (a.double().t() @ b.double()).float()
    err = float((got - want).abs().max()) / float(want.abs().max())
    print(rows, m, n, "max error relative to the largest entry", err)
    assert got.shape == want.shape and err < 2e-6 * max(1.0, (rows / 1000.0) ** 0.5)
    with pytest.raises(ValueError):      # M must be a multiple of 128
        native.gemm_tn_bf16(a[:, :96].contiguous(), b)


def test_eval_loop_miou_parity(wrapper, pkg, oracle, dev):
    """The eval loop of train.py:259-364 / eval.py:60-110 on a small synthetic set (two batches of two
    320 x 320 images): wrapper forward -> UnSegMetrics.update -> compute(), against the CPU oracle's loop on
    the same images.  Cluster mIoU goes through the Hungarian matching, linear mIoU does not.  Weights and
    labels are synthetic, so the mIoU values are small; what is asserted is that both paths give the same."""
    cfg = pkg.configs.S_CFG
    ncls = cfg["num_classes"]
    state = pkg.synth.synth_pqgo_state(cfg, seed=1234)
    batches = [(pkg.synth.synth_images(2, 320, 320, seed=200 + i), pkg.synth.synth_labels(2, 320, 320, ncls, seed=200 + i))
               for i in range(2)]
    want = {}
    conf = {"cluster": torch.zeros(ncls, ncls, dtype=torch.long), "linear": torch.zeros(ncls, ncls, dtype=torch.long)}
    ref_preds = []
    for img, label in batches:
        _t, _o, (o_lin, o_clu), _f, _i = oracle.wrapper_forward(state, img, label, cfg)
        conf["cluster"] += oracle.metrics_confusion(o_clu, label, ncls)
        conf["linear"] += oracle.metrics_confusion(o_lin, label, ncls)
        ref_preds.append((o_lin, o_clu))
    want["cluster"] = float(oracle.metrics_compute(conf["cluster"], True, ncls)[0])
    want["linear"] = float(oracle.metrics_compute(conf["linear"], False, ncls)[0])
    try:
        for precision, miou_tol, agree_floor in (("fp32", 2e-3, 0.9995), ("bf16", 0.05, 0.93)):
            wrapper.model.precision = precision
            cluster_m = pkg.UnSegMetrics(ncls, cfg["eval"]["extra_classes"], True, dev)
            linear_m = pkg.UnSegMetrics(ncls, 0, False, dev)
            agree = []
            with torch.no_grad():
                for (img, label), (o_lin, o_clu) in zip(batches, ref_preds):
                    _total, _out, (lin_p, clu_p), _f = wrapper(img.to(dev), img.to(dev), label.to(dev))
                    cluster_m.update(clu_p, label.to(dev))
                    linear_m.update(lin_p, label.to(dev))
                    agree.append((float((lin_p.cpu() == o_lin).float().mean()), float((clu_p.cpu() == o_clu).float().mean())))
            got_c, got_l = float(cluster_m.compute("cluster")["iou"]), float(linear_m.compute("linear")["iou"])
            print(f"{precision}: cluster mIoU {got_c:.4f} (oracle {want['cluster']:.4f}), linear mIoU {got_l:.4f} "
                  f"(oracle {want['linear']:.4f}), pixel agreement linear/cluster {agree}")
            assert abs(got_c - want["cluster"]) < miou_tol and abs(got_l - want["linear"]) < miou_tol
            assert min(min(a) for a in agree) > agree_floor
    finally:
        wrapper.model.precision = "bf16"
