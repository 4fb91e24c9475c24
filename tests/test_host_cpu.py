"""Host-side logic that needs no GPU: the C-ABI library loads and exports what the header declares,
the module tree reproduces the reference's state_dict, configuration errors surface as the
reference's exception types, and the product path refuses to run without a device."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def _declared(header_name):
    header = open(os.path.join(ROOT, "include", header_name)).read()
    return set(re.findall(r"\b(equss_[a-z0-9_]+)\s*\(", header))


def test_library_exports_every_declared_symbol(native):
    declared = _declared("equss_hip.h")
    assert len(declared) >= 15
    lib = native.load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libequss_hip.so does not export {sym}"
    assert declared == set(native.EXPORTED_SYMBOLS)
    assert lib.equss_abi_version() == 2


def test_library_exports_nothing_undeclared(native):
    """Every equss_* symbol in the dynamic table is declared in include/equss_hip.h (the drop-in ABI) or in
    include/equss_hip_dev.h (developer knobs): nothing that changes behaviour hides outside the headers."""
    import subprocess
    dev = _declared("equss_hip_dev.h")
    assert dev == set(native.DEV_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith("equss_")}
    assert exported == _declared("equss_hip.h") | dev, exported ^ (_declared("equss_hip.h") | dev)


def test_abi_version_agrees_between_header_binding_library_and_build_entry(native):
    """EQUSS_ABI_VERSION of the header, the constant the binding checks, what the library reports, and the driver's
    build() entry point (which asserts the same) all agree."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "equss_hip.h")).read()
    declared = int(re.search(r"#define\s+EQUSS_ABI_VERSION\s+(\d+)", header).group(1))
    assert declared == native.ABI_VERSION == native.load_library().equss_abi_version()
    import __graft_entry__
    __graft_entry__.build()     # up to date: a few seconds; raises if its own checks disagree


def test_workspace_queries_need_no_gpu(native):
    lib = native.load_library()
    assert lib.equss_pq_workspace_bytes(1000, 8, 256, 128) >= 8 * 256 * 128 * 4
    assert lib.equss_pq_workspace_bytes(0, 0, 0, 0) == 0
    assert lib.equss_patch_workspace_bytes(2, 64, 64, 8, 384) > 0


def test_fused_gemm_rejects_operands_beyond_32_bit_offsets(native):
    """The large-tile GEMM kernel addresses its operand pieces by 32-bit element offsets.  Every route into it rejects a
    strided operand whose M * lda (or N * ldw) does not fit, before anything is launched (so this needs no GPU): the fused
    residual + LayerNorm entry point used to reach the kernel unchecked and would have read the wrong rows."""
    import ctypes
    lib = native.load_library()
    fake = ctypes.c_void_p(1 << 20)   # non-null, aligned; never dereferenced: the call fails in argument checking
    rc = lib.equss_gemm_resid_ln_bf16(fake, 65536, fake, 384, fake, fake, 384, fake, fake, fake, 384, 100000, 384, 384,
                                      ctypes.c_float(1e-6), None)
    assert rc == 3 and b"32-bit" in lib.equss_last_error()   # EQUSS_E_UNSUPPORTED
    rc = lib.equss_gemm_bf16(fake, 65536, fake, 384, fake, fake, 1152, 100000, 1152, 384, 0, None)
    assert rc == 3


def test_training_entry_points_check_their_arguments_before_any_launch(native):
    """equss_gemm_tn_bf16, equss_pq_st_backward and equss_probe_train reject bad shapes, alignments and null pointers in
    argument checking (no GPU needed: nothing is dereferenced or launched on these paths)."""
    import ctypes
    lib = native.load_library()
    fake = ctypes.c_void_p(1 << 20)
    assert lib.equss_gemm_tn_bf16(None, 128, fake, 128, fake, 128, 1000, 128, 128, None) == 1          # EQUSS_E_NULL
    assert lib.equss_gemm_tn_bf16(fake, 96, fake, 128, fake, 128, 1000, 96, 128, None) == 2            # M % 128
    assert lib.equss_gemm_tn_bf16(fake, 132, fake, 128, fake, 128, 1000, 128, 128, None) == 4          # lda % 8
    assert lib.equss_gemm_tn_bf16(fake, 1 << 24, fake, 128, fake, 128, 1 << 20, 128, 128, None) == 3   # row range beyond 2 GiB
    f = ctypes.c_float
    assert lib.equss_pq_st_backward(fake, 1024, fake, 1024, fake, None, 0, None, f(0.25), f(1.0), fake, 1024, fake,
                                    100, 8, 256, 128, None) == 1                                       # g_vq is null
    assert lib.equss_pq_st_backward(fake, 1024, fake, 1024, fake, None, 0, fake, f(0.25), f(1.0), fake, 1024, fake,
                                    100, 8, 256, 48, None) == 3                                        # sub-space dim
    assert lib.equss_pq_st_backward(fake, 1000, fake, 1024, fake, None, 0, fake, f(0.25), f(1.0), fake, 1024, fake,
                                    100, 8, 256, 128, None) == 4                                       # ld_z < num_pq * d
    assert lib.equss_probe_train(fake, 1024, fake, fake, fake, fake, fake, fake, None, fake, 1 << 30,
                                 2, 4, 4, 1024, 32, 32, 27, 27, None) == 4                             # no gradient table
    assert b"grad_table" in lib.equss_last_error()


def test_state_dict_matches_reference_layout(pkg):
    for cfg, n_keys in ((pkg.configs.S_CFG, 167), (pkg.configs.B_CFG, 175)):
        m = pkg.build_model(cfg)
        want = pkg.synth.wrapper_state_shapes(cfg)
        got = m.state_dict()
        assert list(got.keys()) == list(want.keys()) and len(got) == n_keys
        for k, shp in want.items():
            assert tuple(got[k].shape) == tuple(shp), k
        m.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=3), strict=True)
        trainable = [p for p in m.parameters() if p.requires_grad]
        assert all(not p.requires_grad for p in m.model.extractor.backbone.parameters())
        assert len(trainable) == (17 if cfg is pkg.configs.S_CFG else 25)
    # the reference's S-cfg trainable parameter count (head 935k + codebooks 262k + probes 55k)
    m = pkg.build_model(pkg.configs.S_CFG)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 1253787


def test_attributes_the_training_harness_touches(pkg):
    m = pkg.build_model(pkg.configs.S_CFG)
    assert hasattr(m, "model") and hasattr(m.evaluator, "cluster_probe") and hasattr(m.evaluator, "linear_probe")
    assert m.model.extractor.n_feats == 384 and m.model.extractor.patch_size == 8
    m.train()
    assert not m.model.extractor.backbone.training      # frozen backbone stays in eval (dino_featurizer.py:58-62)
    assert m.evaluator.cluster_probe.clusters.shape == (27, 1024)


def test_synthetic_state_is_deterministic(pkg):
    a = pkg.synth.synth_pqgo_state(pkg.configs.S_CFG, seed=1234)
    b = pkg.synth.synth_pqgo_state(pkg.configs.S_CFG, seed=1234)
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["model.vq_blocks.0.quantizers.0.embedding.weight"]
    assert float(w.abs().max()) <= 1.0 / 256
    img = pkg.synth.synth_images(1, 16, 16, seed=0)
    assert abs(float(img[0, 0, 0, 0]) - 1.1176220178604126) < 1e-6   # PCG64(0) float32 standard normal, first draw


def test_error_conventions(pkg):
    with pytest.raises(ValueError):                              # dino_pqgo.py:732-733
        pkg.ProductQuantizerWrapper(7, 256, 1024)
    with pytest.raises(ValueError):                              # dino_pqgo.py:91
        pkg.DIONPQGO({**pkg.configs.S_CFG["model"], "vq": {**pkg.configs.S_CFG["model"]["vq"], "vq_type": "bogus"}},
                     pkg.configs.S_CFG["loss"])
    bad = pkg.configs.pqgo_config(output_type="nope")
    with pytest.raises(ValueError):                              # PQGOWrapper.py:44
        pkg.build_model(bad)
    with pytest.raises(ValueError):                              # dino_featurizer.py:36
        pkg.DinoFeaturizer({"model_type": "vit_huge", "dino_patch_size": 8, "pretrained_weights": None})
    with pytest.raises(ValueError):
        pkg.Codebook(32, 64, normalize="bogus")
    with pytest.raises(ValueError):
        pkg.build_model(pkg.configs.S_CFG, name="stego")


def test_no_cpu_fallback(pkg, native):
    m = pkg.build_model(pkg.configs.S_CFG).eval()
    with pytest.raises(native.EqussError):
        m.model(img=torch.zeros(1, 3, 64, 64))
    with pytest.raises(native.EqussError):
        native.pq_assign(torch.zeros(4, 64), torch.zeros(1, 32, 64), "l2")


def test_pos_embed_cache_matches_oracle(pkg, oracle):
    m = pkg.build_model(pkg.configs.S_CFG)
    state = pkg.synth.synth_pqgo_state(pkg.configs.S_CFG, seed=1234)
    m.load_state_dict(state, strict=True)
    bb = m.model.extractor.backbone
    got = bb.interpolate_pos_encoding(6 * 10, 48, 80)
    want = oracle.vit_interpolate_pos_encoding(state["model.extractor.backbone.pos_embed"], 60, 48, 80, 8)
    assert torch.equal(got, want)
    assert bb.interpolate_pos_encoding(28 * 28, 224, 224) is bb.pos_embed


def test_evaluator_matches_oracle_on_cpu(pkg, oracle):
    """The probes run on stock ops, so they can be checked against the oracle without a GPU."""
    cfg = pkg.configs.S_CFG
    state = pkg.synth.synth_pqgo_state(cfg, seed=1234)
    m = pkg.build_model(cfg)
    m.load_state_dict(state, strict=True)
    rng = np.random.Generator(np.random.PCG64(4))
    out = torch.from_numpy(rng.standard_normal((3, 1024, 4, 4)).astype(np.float32))
    label = pkg.synth.synth_labels(3, 32, 32, 27, seed=4)
    with torch.no_grad():
        got = m.evaluator(out, None, label=label)
        want = oracle.evaluator_forward(state, out, label, 27)
    assert torch.allclose(got[0], want[0], rtol=1e-5, atol=1e-6) and torch.equal(got[1], want[1])
    assert torch.allclose(got[2], want[2], rtol=1e-5, atol=1e-6) and torch.equal(got[3], want[3])


def test_bench_work_model_matches_survey_numbers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import importlib
    pkg = importlib.import_module("expand-and-quantize-for-unsupervised-semantic-segmentation_amd")
    w = bench.algorithmic_work(pkg.configs.S_CFG, 1, 320, 320)
    block = sum(w[k]["flops"] for k in ("gemm_qkv", "attention", "gemm_proj", "gemm_fc1", "gemm_fc2"))
    assert abs(block / 1e9 - 9.603) < 0.02                        # GFLOP per block per image
    assert abs((w["head_gemm1"]["flops"] + w["head_gemm2"]["flops"]) / 1e9 - 2.99) < 0.01
    assert abs(w["pq_assign"]["flops"] / 1e9 - 0.839) < 0.001
    assert abs(w["pq_assign"]["bytes"] / 1e6 - 13.21) < 0.01       # MB per image


def test_point_sampling_matches_grid_sample(pkg):
    """The four-neighbour gather of the STEGO loss against F.grid_sample (loss.py:647-648), values and
    gradients, including coordinates on the border and permuted images."""
    import torch.nn.functional as F
    pq = pkg.pqgo
    torch.manual_seed(3)
    b, c, h, w, n = 3, 5, 6, 7, 4
    t = torch.randn(b, h * w, c).requires_grad_(True)
    tm = t.reshape(b, h, w, c).permute(0, 3, 1, 2)                      # channels-first view of token-major rows
    coords = torch.rand(2, b, n, n, 2) * 2 - 1
    coords[0, 0, 0, 0] = torch.tensor([1.0, -1.0])
    coords[0, 0, 0, 1] = torch.tensor([-1.0, 1.0])
    coords[1, 1, 2, 2] = torch.tensor([1.0, 1.0])
    perm = torch.tensor([2, 0, 1])
    which = torch.stack([torch.arange(b), perm])
    got = pq._sample_points(tm, coords, which)
    ref0 = F.grid_sample(tm, coords[0].permute(0, 2, 1, 3), padding_mode="border", align_corners=True)
    ref1 = F.grid_sample(tm[perm], coords[1].permute(0, 2, 1, 3), padding_mode="border", align_corners=True)
    torch.testing.assert_close(got[0], ref0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got[1], ref1, rtol=1e-5, atol=1e-6)
    wgt = torch.randn_like(got)
    (g_got,) = torch.autograd.grad((got * wgt).sum(), t)
    (g_ref,) = torch.autograd.grad((ref0 * wgt[0]).sum() + (ref1 * wgt[1]).sum(), t)
    torch.testing.assert_close(g_got, g_ref, rtol=1e-5, atol=1e-6)


def test_stego_loss_replay_matches_oracle(pkg, oracle):
    """The package's STEGOLoss (one batched four-neighbour gather per tensor) with replayed draws against
    the oracle's restatement of loss.py:701-739 (grid_sample per point set), value and gradient, on CPU."""
    torch.manual_seed(3)
    cfg = pkg.configs.S_CFG["loss"]["stego"]
    b, c, d, h, w, n = 3, 24, 40, 6, 7, cfg["feature_samples"]
    feats, feats_pos = torch.randn(b, c, h, w), torch.randn(b, c, h, w)
    code = torch.randn(b, d, h, w, requires_grad=True)
    code_pos = torch.randn(b, d, h, w, requires_grad=True)
    coords1, coords2 = torch.rand(b, n, n, 2) * 2 - 1, torch.rand(b, n, n, 2) * 2 - 1
    perms = [pkg.pqgo._derangement_like(b, torch.device("cpu")) for _ in range(cfg["neg_samples"])]
    want = oracle.stego_loss(cfg, feats, feats_pos, code, code_pos, coords1, coords2, perms)
    g_want = torch.autograd.grad(want, (code, code_pos))
    loss_mod = pkg.pqgo.STEGOLoss(cfg)
    loss_mod.replay = (coords1, coords2, perms)
    got = loss_mod(feats, feats_pos, code, code_pos)
    g_got = torch.autograd.grad(got, (code, code_pos))
    assert abs(float(got) - float(want)) < 1e-6 + 1e-5 * abs(float(want))
    for a, e in zip(g_got, g_want):
        assert float((a - e).abs().max()) < 1e-6 + 1e-4 * float(e.abs().max())
    loss_mod.replay = (coords1[:2], coords2[:2], perms)
    with pytest.raises(ValueError):
        loss_mod(feats, feats_pos, code, code_pos)


def test_code_index_dump_round_trip(pkg, tmp_path):
    """The index dump (train.py:283-329 sketch): nearest-upsampled indices next to labels, appended
    batch by batch, readable with numpy.load; against F.interpolate(mode="nearest")."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    num_pq, h, w, hh, ww = 3, 5, 4, 40, 32
    prefix = str(tmp_path / "dump" / "val")
    want_idx, want_lab = [], []
    with pkg.CodeIndexDump(prefix, num_pq) as dump:
        for b in (2, 1):
            z = torch.randint(0, 512, (num_pq, b, h, w))
            lab = torch.randint(-1, 27, (b, hh, ww))
            dump.append(z, lab)
            up = F.interpolate(z.permute(1, 0, 2, 3).float(), size=(hh, ww), mode="nearest")     # (b, num_pq, H, W)
            want_idx.append(up.reshape(b, num_pq, -1).permute(0, 2, 1).reshape(-1, num_pq).numpy())
            want_lab.append(lab.reshape(-1).numpy())
    idx = np.load(prefix + "_index.npy", mmap_mode="r")
    lab = np.load(prefix + "_label.npy")
    assert idx.dtype == np.int16 and idx.shape == (3 * hh * ww, num_pq) and lab.shape == (3 * hh * ww,)
    assert np.array_equal(idx, np.concatenate(want_idx).astype(np.int16))
    assert np.array_equal(lab, np.concatenate(want_lab).astype(np.int16))
    with pkg.CodeIndexDump(prefix + "_low", num_pq, upsample=False) as dump:
        z = torch.randint(0, 512, (num_pq, 2, h, w))
        dump.append(z, torch.zeros(2, hh, ww, dtype=torch.long))
    assert np.load(prefix + "_low_index.npy").shape == (2 * h * w, num_pq)


def test_no_register_is_used_while_an_lds_read_into_it_is_pending(pkg, tmp_path):
    """The ring kernels issue their fragment reads as inline-asm ds_read_b128 and wait for them with counted
    s_waitcnt lgkmcnt(N).  The compiler takes such a read's result for valid from the instruction on and, with all
    registers in use, may copy the register (to an AccVGPR, to scratch) before the data is there: garbage results, no
    fault.  tools/lds_hazard_check.py scans the gfx950 assembly of every kernel file that reads LDS this way, per
    basic block, for any use of a register with a read pending (the scanner itself is checked on a two-line example)."""
    import importlib.util
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    spec = importlib.util.spec_from_file_location("lds_hazard_check", os.path.join(ROOT, "tools", "lds_hazard_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    bad = tmp_path / "bad.s"
    bad.write_text("\tds_read_b128 v[8:11], v1 offset:0\n\tv_accvgpr_write_b32 a3, v11\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v0, v8\n")
    assert chk.scan(str(bad)) == 1
    # a scalar load pending on the same counter at a counted wait behind an inline-asm read, and such a read left pending
    # where its block ends: each flagged once; the same with reads the compiler placed itself: not flagged
    asm_read = "\t;;#ASMSTART\n\tds_read_b128 v[8:11], v1 offset:0\n\t;;#ASMEND\n"
    bad.write_text(asm_read + "\tds_read_b128 v[12:15], v1 offset:16\n\ts_load_dword s4, s[0:1], 0x0\n\ts_waitcnt lgkmcnt(1)\n"
                   "\tv_mov_b32_e32 v0, v8\n\ts_waitcnt lgkmcnt(0)\n")
    assert chk.scan(str(bad)) == 1
    bad.write_text(asm_read + "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt lgkmcnt(0)\n")
    assert chk.scan(str(bad)) == 1
    bad.write_text("\tds_read_b128 v[8:11], v1 offset:0\n\ts_load_dword s4, s[0:1], 0x0\n\ts_cbranch_scc1 .LBB0_1\n")
    assert chk.scan(str(bad)) == 0
    import importlib
    build = importlib.import_module(pkg.__name__ + ".build")
    if not os.path.exists(build.HIPCC):
        pytest.skip("no hipcc here")
    files = ["block_ws.hip", "mlp_fused.hip", "gemm_ws.hip", "gemm8_bf16.hip", "attention.hip", "block_rs.hip", "gemm_tn.hip"]

    def flags_of(name):
        out = tmp_path / (name[:-4] + ".s")
        cmd = [build.HIPCC] + build.FLAGS + build.FILE_FLAGS.get(name, []) + ["-S", "--cuda-device-only", "-o", str(out),
                                                                              os.path.join(build.CSRC, name)]
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert res.returncode == 0, res.stdout
        return chk.scan(str(out))

    with ThreadPoolExecutor(max_workers=4) as ex:
        flagged = dict(zip(files, ex.map(flags_of, files)))
    assert flagged == {f: 0 for f in files}
