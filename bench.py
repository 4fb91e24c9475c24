"""Benchmark of the EQUSS hot path on MI355X: images/s at 320x320 for DINO-ViT patch features ->
expansion head -> product-quantisation assign (BASELINE.json's metric), one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one eval forward of DIONPQGO (ViT-S/8, D=1024, 8 sub-spaces x 256 codewords) over one
synthetic batch already resident in HBM.  Images are independent, so ranks shard them with no
data-path collective (weak scaling: the per-GPU batch is fixed).  Rank 0 prints one JSON line.

Besides the contract fields the line carries
  roofline      - the kernel class with the largest share of the step, priced against the dense
                  bf16 MFMA peak, from HIP events recorded around every launch of a second pass of
                  the same K steps (the timed pass itself runs without events);
  roofline_pq   - the PQ-assign kernel against HBM bandwidth;
  kernels       - every kernel class: ms per step, launches, achieved rate;
  cpu_baseline  - the CPU oracle (same algorithm, fp32, PyTorch CPU ops) timed on this node's host
                  cores on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = "expand-and-quantize-for-unsupervised-semantic-segmentation_amd"
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA = 2.5e15     # dense, FLOP/s
PEAK_HBM = 8.0e12           # B/s
PEAK_FP32 = 157.3e12        # FLOP/s (vector = f32 MFMA rate)


def algorithmic_work(cfg, batch, height, width, block_ws=True):
    """FLOPs (2 M N K, no recompute) and bytes per LAUNCH of each kernel class for one batch."""
    from_cfg = cfg["model"]
    c = {"vit_small": 384, "vit_base": 768}[from_cfg["pretrained"]["model_type"]]
    p = from_cfg["pretrained"]["dino_patch_size"]
    d_out = from_cfg["vq"]["embed_dims"][0]
    k = from_cfg["vq"]["num_codebooks"][0]
    num_pq = from_cfg["vq"]["num_pq"][0]
    n = (height // p) * (width // p)
    tokens = n + 1
    rows = batch * tokens
    # ViT-S from 384 row tiles up: one fused launch per block covers the whole rounds of 128-row tiles on
    # the 256 CUs (proj + MLP), the trailing rows go through one small launch per GEMM (vit_ops.hip)
    tiles = (rows + 127) // 128
    fused = c == 384 and tiles >= 384
    # (the default producer / consumer block-tail kernel takes all rows; the eight-wave one, --dev block_ws=0, only the
    # whole rounds of 128-row tiles)
    split = fused and not block_ws and tiles // 256 >= 1 and 0 < tiles % 256 <= 128
    rows_fused = (tiles // 256) * 256 * 128 if split else (rows if fused else 0)
    rows_gemm = rows - rows_fused          # rows the per-GEMM launches of proj / fc1 / fc2 see
    w = {
        "patch": {"flops": 2.0 * batch * n * (3 * p * p) * c},
        "layernorm": {"bytes": rows * c * (4 + 2)},
        "gemm_qkv": {"flops": 2.0 * rows * c * 3 * c},
        "attention": {"flops": 4.0 * batch * tokens * tokens * c},
        "gemm_proj": {"flops": 2.0 * rows_gemm * c * c},
        "gemm_fc1": {"flops": 2.0 * rows_gemm * c * 4 * c},
        "gemm_fc2": {"flops": 2.0 * rows_gemm * 4 * c * c},
        "mlp_fused": {"flops": 4.0 * rows * 4 * c * c},     # fc1 + fc2 in one launch (ViT-S)
        "block_tail": {"flops": 2.0 * rows_fused * c * c + 4.0 * rows_fused * 4 * c * c},   # proj + fc1 + fc2, one launch
        "head_gemm1": {"flops": 2.0 * batch * n * c * c},
        "head_gemm2": {"flops": 2.0 * batch * n * 2 * c * d_out},
        "pq_prep": {"bytes": 2 * num_pq * k * (d_out // num_pq) * 4},
        # read code + write z_q + write int64 indices; codebooks are L2/register resident
        "pq_assign": {"bytes": batch * n * (2 * d_out * 4 + num_pq * 8), "flops": 2.0 * batch * n * d_out * k},
    }
    return w


def _timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def secondary_workloads(pkg, dev, res):
    """The other configurations of the metric's family, each on a handful of steps after the timed region (bounded:
    about ten seconds in all): ViT-B/8 + 16 x 512 PQ at 768 x 768, the eval loop with probes and metrics, one training step."""
    out = {}
    # ---- ViT-B/8, D = 2048, 16 x 512 PQ, 2 x 768 x 768 per GPU ------------------------------------------------------
    cfg_b = pkg.configs.B_CFG
    model_b = pkg.build_model(cfg_b)
    model_b.load_state_dict(pkg.synth.synth_pqgo_state(cfg_b, seed=1234), strict=True)
    model_b = model_b.to(dev).eval()
    img_b = pkg.synth.synth_images(2, 768, 768, seed=7).to(dev)

    def step_b():
        with torch.no_grad():
            model_b.model(img=img_b)
    s_b = _timed(step_b, 2, 4)
    pkg.native.profile_enable(4096)
    step_b()
    prof = pkg.native.profile_read(reset=True)
    pkg.native.profile_enable(0)
    work_b = algorithmic_work(cfg_b, 2, 768, 768)
    pq_ms, pq_n = prof["pq_assign"]
    flops_b = sum(work_b[k]["flops"] * prof[k][1] for k in work_b if "flops" in work_b[k] and k in prof and prof[k][1])
    out["vit_b8_768"] = {"workload": "ViT-B/8 + head (D=2048) + 16 x 512 PQ, 2 x 768x768, eval forward",
                         "images_per_s": round(2 / s_b, 2), "ms_per_step": round(s_b * 1e3, 3),
                         "model_tflops": round(flops_b / s_b / 1e12, 1),
                         "pq_assign_us": round(pq_ms / max(pq_n, 1) * 1e3, 1),
                         "pq_assign_frac_of_hbm_peak": round(work_b["pq_assign"]["bytes"] / (pq_ms / max(pq_n, 1) / 1e3) / PEAK_HBM, 4)}
    del model_b, img_b
    torch.cuda.empty_cache()
    # ---- eval loop: wrapper forward (ViT + head + PQ + probes) + confusion-matrix updates + compute() ------------------
    cfg = pkg.configs.S_CFG
    ncls = cfg["num_classes"]
    state = pkg.synth.synth_pqgo_state(cfg, seed=1234)
    wrapper = pkg.build_model(cfg)
    wrapper.load_state_dict(state, strict=True)
    wrapper = wrapper.to(dev).eval()
    cluster_m = pkg.UnSegMetrics(ncls, cfg["eval"]["extra_classes"], True, dev)
    linear_m = pkg.UnSegMetrics(ncls, 0, False, dev)
    pool = [(pkg.synth.synth_images(64, res, res, seed=100 + i).to(dev),
             pkg.synth.synth_labels(64, res, res, ncls, seed=100 + i).to(dev)) for i in range(2)]

    def eval_loop(n):
        cluster_m.reset()
        linear_m.reset()
        with torch.no_grad():
            for i in range(n):
                img, label = pool[i % len(pool)]
                _t, _o, (lin_p, clu_p), _f = wrapper(img, img, label)
                cluster_m.update(clu_p, label)
                linear_m.update(lin_p, label)
        return cluster_m.compute("cluster"), linear_m.compute("linear")
    eval_loop(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_eval = 12
    res_c, res_l = eval_loop(n_eval)
    torch.cuda.synchronize()
    s_e = time.perf_counter() - t0
    out["eval_loop"] = {"workload": f"eval loop, {n_eval} batches of 64 x {res}x{res}: wrapper forward (ViT-S/8 + head + PQ + probes) + "
                                    "metric updates + compute()",
                        "images_per_s": round(n_eval * 64 / s_e, 1), "cluster_miou": round(float(res_c["iou"]), 4),
                        "linear_miou": round(float(res_l["iou"]), 4)}
    del wrapper, pool
    torch.cuda.empty_cache()
    # ---- one training step: 16 x 224 x 224 (config/pqgo_baseline.yaml:135,146), train.py:74-98 ------------------------
    cfg_t = pkg.configs.pqgo_config(need_initialized="uni")
    trainee = pkg.build_model(cfg_t)
    trainee.load_state_dict(pkg.synth.synth_pqgo_state(cfg_t, seed=1234), strict=True)
    trainee = trainee.to(dev).train()
    optim = torch.optim.Adam([p for p in trainee.parameters() if p.requires_grad], lr=3e-4)
    img_t = pkg.synth.synth_images(16, 224, 224, seed=1).to(dev)
    pos_t = pkg.synth.synth_images(16, 224, 224, seed=101).to(dev)
    lab_t = pkg.synth.synth_labels(16, 224, 224, cfg_t["num_classes"], seed=1).to(dev)
    it = [0]

    def train_step():
        optim.zero_grad(set_to_none=True)
        with torch.enable_grad():   # (the CPU-baseline leg switches autograd off globally)
            total, _o, _p, _f = trainee(img=img_t, aug_img=img_t, label=lab_t, img_pos=pos_t, it=it[0])
            total.backward()
        torch.nn.utils.clip_grad_norm_(trainee.model.parameters(), max_norm=10.0)
        optim.step()
        it[0] += 1
    s_t = _timed(train_step, 3, 8)
    out["train_step"] = {"workload": "training step, 16 x 224x224: frozen ViT-S/8 on image + positive pair, head / codebooks / probes forward, "
                                     "backward, clip, Adam", "ms_per_step": round(s_t * 1e3, 3),
                         "images_per_s": round(16 / s_t, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--res", type=int, default=320)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the batch's image ranges are spread over inside one step (1 = one stream)")
    ap.add_argument("--dev", action="append", default=[], metavar="KNOB=INT",
                    help="developer tuning knob of the library (equss_dev_KNOB), for experiments only")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the fp32-mode and secondary-workload legs (B-cfg 768, eval loop, training step) after the timed region")
    ap.add_argument("--cfg", choices=["s", "b"], default="s",
                    help="s = ViT-S/8, D=1024, 8x256 PQ (the headline workload); b = ViT-B/8, D=2048, 16x512 PQ "
                         "(use with --res 768 --batch 2..4)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the EQUSS path has no CPU fallback")
    # EQUSS_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend, to walk the N > 1 code path on a one-GPU box
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
    rehearsal = os.environ.get("EQUSS_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if rehearsal else "nccl")  # nccl = RCCL over xGMI on ROCm

    pkg = importlib.import_module(PKG)
    cfg = pkg.configs.S_CFG if args.cfg == "s" else pkg.configs.B_CFG
    state = pkg.synth.synth_pqgo_state(cfg, seed=1234)
    model = pkg.build_model(cfg)
    model.load_state_dict(state, strict=True)
    model = model.to(dev).eval()
    model.model.eval_streams = max(1, args.streams)
    for knob in args.dev:
        name, value = knob.split("=")
        getattr(pkg.native.load_library(), "equss_dev_" + name)(int(value))
    img = pkg.synth.synth_images(args.batch, args.res, args.res, seed=1 + rank).to(dev)

    def step():
        with torch.no_grad():
            return model.model(img=img)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps, no events, no host sync inside ------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    z_index = out[2][2]
    assert z_index.shape[0] == cfg["model"]["vq"]["num_pq"][0] and z_index.dtype == torch.int64

    # ---- same K steps again with a HIP event pair around every kernel-class launch ---------------
    # (on ONE stream: with the batch spread over several streams kernels of different image ranges share
    # the device, and a kernel's own duration could not be told from its neighbours')
    model.model.eval_streams = 1
    per_step_launches = 12 * 7 + 8
    pkg.native.profile_enable(per_step_launches * args.steps + 64)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    profiled = time.perf_counter() - t1
    prof = pkg.native.profile_read(reset=True)
    pkg.native.profile_enable(0)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    knobs = dict(kv.split("=") for kv in args.dev)
    work = algorithmic_work(cfg, args.batch, args.res, args.res,
                            block_ws=int(knobs.get("block_ws", 3)) != 0 and int(knobs.get("tail_images", 1)) != 0)
    # HBM bytes per launch from the PMC counters are measured by separate rocprofv3 passes (profiles/): they
    # are quoted only while the kernel sources still hash to what those passes ran, otherwise null
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f)
        digest = importlib.import_module(PKG + ".build").source_digest()
        if traffic.get("source_digest") != digest:
            traffic = {}
    kernels = {}
    share = 1.0   # the profiled pass launches every kernel once over the whole batch
    for name, (ms, launches) in prof.items():
        if launches == 0 or name not in work:
            continue
        avg_s = ms / launches / 1e3
        entry = {"ms_per_step": round(ms / args.steps, 4), "launches_per_step": launches / args.steps,
                 "avg_launch_us": round(avg_s * 1e6, 2)}
        if "flops" in work[name]:
            entry["tflops"] = round(share * work[name]["flops"] / avg_s / 1e12, 2)
        if "bytes" in work[name]:
            entry["gbps"] = round(share * work[name]["bytes"] / avg_s / 1e9, 1)
        kernels[name] = entry
    dominant = max((k for k in kernels if "tflops" in kernels[k] and k != "pq_assign"),
                   key=lambda k: kernels[k]["ms_per_step"])
    dom = kernels[dominant]
    roofline = {"kernel": dominant, "bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_BF16_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": round(dom["tflops"] * 1e12 / PEAK_BF16_MFMA, 4),
                "traffic": traffic.get(dominant), "avg_launch_us": dom["avg_launch_us"],
                "share_of_step": round(dom["ms_per_step"] / sum(k["ms_per_step"] for k in kernels.values()), 3),
                "profiled_ms_per_step": round(profiled / args.steps * 1e3, 3),
                "measured": "HIP event pairs around every kernel launch in a second pass of the same K steps on one stream "
                            "(the timed pass runs on config.streams_per_gpu stream(s))"}
    pq = kernels["pq_assign"]
    roofline_pq = {"kernel": "pq_assign", "bound": "hbm", "achieved": pq["gbps"], "peak": PEAK_HBM / 1e9,
                   "unit": "GB/s", "frac": round(pq["gbps"] * 1e9 / PEAK_HBM, 4), "traffic": traffic.get("pq_assign"),
                   "avg_launch_us": pq["avg_launch_us"],
                   # 2 n D K flop / time: what an exhaustive fp32 search would have to sustain (fp32 matrix peak
                   # 157 TFLOP/s); the kernel screens with fp16 MFMA and re-scores only the survivors in fp32
                   "exhaustive_fp32_equiv_tflops": pq["tflops"]}
    total_flops = sum(share * work[k]["flops"] * prof[k][1] / args.steps for k in work if "flops" in work[k] and prof[k][1])
    ms_per_step = elapsed / args.steps * 1e3

    cpu_baseline = None
    index_match = None
    if world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        oracle = importlib.import_module("equss_oracle")
        cpu_img = pkg.synth.synth_images(4, args.res, args.res, seed=0)
        torch.set_grad_enabled(False)
        oracle.pqgo_forward(state, cpu_img, cfg)  # warm-up
        best, runs, spent = float("inf"), 0, 0.0
        while runs < 2 or (spent < args.cpu_seconds and runs < 8):
            c0 = time.perf_counter()
            oracle.pqgo_forward(state, cpu_img, cfg)
            dt = time.perf_counter() - c0
            best, runs, spent = min(best, dt), runs + 1, spent + dt
        # second half of the metric: code-index match of the GPU path on the same 4 images
        with torch.no_grad():
            g_code, _q, (_s, _z, g_index), _o = model.model(img=cpu_img.to(dev))
        _c, _qq, ref_index, _oo = oracle.pqgo_forward(state, cpu_img, cfg)
        num_pq = cfg["model"]["vq"]["num_pq"][0]
        books = [state[f"model.vq_blocks.0.quantizers.{i}.embedding.weight"] for i in range(num_pq)]
        own_rows = g_code.permute(0, 2, 3, 1).reshape(-1, g_code.shape[1]).cpu().contiguous()
        want, gap = oracle.pq_top2_gap(books, own_rows, cfg["model"]["vq"].get("normalize", "none"))
        got = g_index.reshape(num_pq, -1).cpu()
        model.model.precision = "fp32"      # parity mode: the whole forward in fp32 on the GPU
        with torch.no_grad():
            _c32, _q32, (_s32, _z32, p_index), _o32 = model.model(img=cpu_img.to(dev))
        model.model.precision = "bf16"
        index_match = {
            # the PQ stage alone: GPU indices vs the fp64 arg-min of the SAME fp32 code rows
            "own_code_exact": round(float((got == want).float().mean()), 6),
            "own_code_mismatch_outside_near_ties": int(((got != want) & (gap > 2e-6)).sum()),
            # whole path: bf16 ViT/head + fp32 PQ vs the fp32 CPU reference restatement
            "e2e_vs_fp32_reference": round(float((g_index.cpu() == ref_index).float().mean()), 6),
            # whole path in the fp32 parity mode (DIONPQGO.precision = "fp32") vs the same reference
            "e2e_fp32_mode_vs_fp32_reference": round(float((p_index.cpu() == ref_index).float().mean()), 6),
            "assignments": int(got.numel()), "sample": f"4 images {args.res}x{args.res}, seed 0"}
        cpu_baseline = {"value": round(4.0 / best, 3), "unit": "images/s", "cores": torch.get_num_threads(),
                        "kind": "port",
                        "sample": f"model forward (ViT+head+PQ) on 4 images {args.res}x{args.res}, fp32, "
                                  f"best of {runs} runs after 1 warm-up, {spent:.1f} s of CPU time"}

    # ---- the mode whose code indices equal the reference's (index_match.e2e_fp32_mode_vs_fp32_reference), at the same batch ----
    fp32_mode = None
    secondary = None
    if world == 1 and not args.no_secondary:
        model.model.precision = "fp32"
        model.model.eval_streams = 1
        s_f = _timed(step, 1, 3)
        model.model.precision = "bf16"
        fp32_mode = {"images_per_s": round(args.batch / s_f, 2), "ms_per_step": round(s_f * 1e3, 3),
                     "what": "the same forward with DIONPQGO.precision = 'fp32' (every GEMM and the attention in fp32 on the "
                             "matrix cores): the mode whose code indices equal the fp32 reference's",
                     "vs_cpu_baseline": round(args.batch / s_f / cpu_baseline["value"], 1) if cpu_baseline else None}
        if args.cfg == "s":
            del model
            torch.cuda.empty_cache()
            secondary = secondary_workloads(pkg, dev, args.res)

    line = {
        "metric": "images/sec (320x320) end-to-end fwd+PQ-assign; PQ code-index exact-match vs ref",
        "value": round(world * args.batch * args.steps / elapsed, 2),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"DINO {'ViT-S/8' if args.cfg == 's' else 'ViT-B/8'} + expansion head "
                               f"(D={cfg['model']['vq']['embed_dims'][0]}) + {cfg['model']['vq']['num_pq'][0]}-subspace PQ "
                               f"(K={cfg['model']['vq']['num_codebooks'][0]}), "
                               f"batch {args.batch} x {args.res}x{args.res} per GPU, bf16 ViT/head + fp32 PQ, eval forward",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "resolution": args.res,
                   "parallelism": f"dp{world} (images sharded, no data-path collective)",
                   "streams_per_gpu": max(1, min(args.streams, args.batch))},
        "model_tflops": round(total_flops / (elapsed / args.steps) / 1e12, 1),
        "roofline": roofline,
        "roofline_pq": roofline_pq,
        "kernels": kernels,
        "cpu_baseline": cpu_baseline,
        "index_match": index_match,
        "fp32_mode": fp32_mode,
        "secondary": secondary,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
