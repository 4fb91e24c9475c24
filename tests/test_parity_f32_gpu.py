"""fp32 parity mode (DIONPQGO.precision = "fp32"): the whole eval forward in fp32 on the GPU against the
reference's fp32 CPU outputs frozen in tests/golden.  What separates the two runs is only the summation
order of fp32 dot products, so features and `code` agree to ~1e-6 relative and the code indices agree
everywhere except where the reference's own top-2 distance gap (fp64, frozen with the fixture) is of
that size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP32_REL_L2 = 2e-5       # 12 blocks of fp32 GEMMs in a different summation order: measured ~1e-6
INDEX_TIE_GAP = 1e-5     # an index may differ only where the reference's fp64 top-2 gap is below this (measured: none differ)


def _rel_l2(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want) / np.linalg.norm(want))


def _model(pkg, dev, cfg):
    m = pkg.build_model(cfg)
    m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    m = m.to(dev).eval()
    m.model.precision = "fp32"
    return m


def test_fp32_vit_matches_reference_taps_and_cls(pkg, golden, dev):
    g = golden("g1_vit_64")
    m = _model(pkg, dev, pkg.configs.S_CFG)
    img = pkg.synth.synth_images(2, 64, 64, seed=11).to(dev)
    feats, _, _ = m.model.extractor.backbone.get_intermediate_feat(img, n=1)
    err = _rel_l2(feats[0].cpu().numpy(), g["feat"])
    print("fp32 mode, G1 features incl. CLS rel-L2:", err, "max abs:", float(np.abs(feats[0].cpu().numpy() - g["feat"]).max()))
    assert err < FP32_REL_L2
    feat = m.model.extractor(img)
    assert feat.shape == (2, 384, 8, 8)
    assert _rel_l2(feat.permute(0, 2, 3, 1).reshape(2, 64, 384).cpu().numpy(), g["feat"][:, 1:]) < FP32_REL_L2


def test_fp32_mode_indices_equal_reference_outside_near_ties(pkg, golden, dev):
    """S-cfg, 4 x 320 x 320 (fixture G5): 51 200 assignments."""
    g = golden("g5_scfg_b4_320")
    cfg = pkg.configs.S_CFG
    m = _model(pkg, dev, cfg)
    img = pkg.synth.synth_images(4, 320, 320, seed=0).to(dev)
    label = pkg.synth.synth_labels(4, 320, 320, cfg["num_classes"], seed=0).to(dev)
    with torch.no_grad():
        code, quant, (_s, _z, z_index), outputs = m.model(img=img)
        total, out, (lin_p, clu_p), feat_vqs = m(img, img, label)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])   # noqa: E731
    tok = g["tok"]
    e_feat = _rel_l2(rows(m.model.extractor(img))[tok].cpu().numpy(), g["feat_tok"])
    e_code = _rel_l2(rows(code)[tok].cpu().numpy(), g["code_tok"])
    got, want, gap = z_index.cpu().numpy(), g["index"].astype(np.int64), g["gap64"]
    diff = got != want
    print(f"fp32 mode, G5: feature rel-L2 {e_feat:.2e}, code rel-L2 {e_code:.2e}; {int(diff.sum())} of {diff.size} indices differ, "
          f"largest reference gap among them {float(gap[diff].max()) if diff.any() else 0.0:.2e}; "
          f"assignments with gap below {INDEX_TIE_GAP}: {int((gap < INDEX_TIE_GAP).sum())}")
    assert e_feat < FP32_REL_L2 and e_code < FP32_REL_L2
    assert not (diff & (gap > INDEX_TIE_GAP)).any()
    assert diff.sum() <= (gap < INDEX_TIE_GAP).sum() and diff.mean() < 1e-3
    # quantised rows, losses and probe predictions follow
    q = rows(quant)[tok].cpu().numpy().reshape(len(tok), 8, 128)
    same = (got.reshape(8, -1)[:, tok] == want.reshape(8, -1)[:, tok]).T
    np.testing.assert_allclose(q[same], g["quant_tok"].reshape(len(tok), 8, 128)[same], rtol=0, atol=3e-7)
    assert abs(float(out["vq-loss"]) - float(g["vq_loss"])) < 1e-4 * float(g["vq_loss"])
    assert abs(float(out["linear-loss"]) - float(g["linear_loss"])) < 1e-4 * abs(float(g["linear_loss"]))
    assert abs(float(out["cluster-loss"]) - float(g["cluster_loss"])) < 1e-4 * abs(float(g["cluster_loss"]))
    pix = g["pix"]
    assert (lin_p.reshape(-1)[pix].cpu().numpy() == g["linear_preds"]).mean() > 0.999
    assert (clu_p.reshape(-1)[pix].cpu().numpy() == g["cluster_preds"]).mean() > 0.999


def test_fp32_mode_bcfg_indices(pkg, golden, dev):
    """ViT-B/8, D = 2048, 16 x 512 codewords on the 64 x 96 fixture (G11)."""
    g = golden("g11_bcfg_64x96")
    m = _model(pkg, dev, pkg.configs.B_CFG)
    img = pkg.synth.synth_images(1, 64, 96, seed=13).to(dev)
    with torch.no_grad():
        code, _q, (_s, _z, z_index), outputs = m.model(img=img)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])   # noqa: E731
    assert _rel_l2(rows(code).cpu().numpy(), g["code_rows"]) < FP32_REL_L2
    got, want, gap = z_index.reshape(16, -1).cpu().numpy(), g["index"].reshape(16, -1).astype(np.int64), g["gap64"]
    diff = got != want
    assert not (diff & (gap > INDEX_TIE_GAP)).any(), int((diff & (gap > INDEX_TIE_GAP)).sum())
    assert abs(float(outputs["vq-loss"]) - float(g["vq_loss"])) < 1e-4 * float(g["vq_loss"])


def test_precision_switch_is_validated(pkg, dev):
    m = pkg.build_model(pkg.configs.S_CFG)
    assert m.model.precision == "bf16"
    with pytest.raises(ValueError):
        m.model.precision = "fp16"
