"""Eval loop of the reference's harness (train.py:259-364 with resume.checkpoint set; eval.py:60-110)
on a synthetic COCO-Stuff-27-shaped set: wrapper forward -> UnSegMetrics.update per batch ->
compute() at the end.  Prints one JSON line: images/s of the whole loop, the model forward's share,
cluster / linear mIoU, and (with --check K) the same metrics from the CPU oracle on the first K
images with the confusion-matrix difference.

  python tools/eval_loop.py --images 512 --batch 64 --check 4
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "expand-and-quantize-for-unsupervised-semantic-segmentation_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--res", type=int, default=320)
    ap.add_argument("--check", type=int, default=0, help="oracle-check the first K images on the CPU")
    ap.add_argument("--stock-probes", action="store_true", help="time the stock-op probe path instead of the HIP one")
    args = ap.parse_args()
    pkg = importlib.import_module(PKG)
    dev = torch.device("cuda", 0)
    cfg = pkg.configs.S_CFG
    ncls = cfg["num_classes"]
    state = pkg.synth.synth_pqgo_state(cfg, seed=1234)
    model = pkg.build_model(cfg)
    model.load_state_dict(state, strict=True)
    model = model.to(dev).eval()
    model.evaluator.use_hip = not args.stock_probes
    cluster_m = pkg.UnSegMetrics(ncls, cfg["eval"]["extra_classes"], True, dev)
    linear_m = pkg.UnSegMetrics(ncls, 0, False, dev)
    n_batches = (args.images + args.batch - 1) // args.batch
    # the "dataset": a few distinct resident batches, cycled
    pool = [(pkg.synth.synth_images(args.batch, args.res, args.res, seed=100 + i).to(dev),
             pkg.synth.synth_labels(args.batch, args.res, args.res, ncls, seed=100 + i).to(dev)) for i in range(min(4, n_batches))]

    def run(n):
        cluster_m.reset()
        linear_m.reset()
        with torch.no_grad():
            for i in range(n):
                img, label = pool[i % len(pool)]
                _total, _out, (lin_p, clu_p), _f = model(img, img, label)
                cluster_m.update(clu_p, label)
                linear_m.update(lin_p, label)
        return cluster_m.compute("cluster"), linear_m.compute("linear")

    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res_c, res_l = run(n_batches)
    torch.cuda.synchronize()
    loop_s = time.perf_counter() - t0
    with torch.no_grad():
        t1 = time.perf_counter()
        for i in range(n_batches):
            model.model(img=pool[i % len(pool)][0])
        torch.cuda.synchronize()
        model_s = time.perf_counter() - t1
    line = {"workload": f"eval loop, ViT-S/8 + PQ + probes + metrics, {n_batches} batches of {args.batch} x {args.res}^2, synthetic",
            "probes": "stock ops" if args.stock_probes else "hip",
            "images_per_s": round(n_batches * args.batch / loop_s, 1),
            "model_forward_only_images_per_s": round(n_batches * args.batch / model_s, 1),
            "cluster_miou": round(float(res_c["iou"]), 4), "cluster_acc": round(float(res_c["accuracy"]), 4),
            "linear_miou": round(float(res_l["iou"]), 4), "linear_acc": round(float(res_l["accuracy"]), 4)}

    if args.check > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        oracle = importlib.import_module("equss_oracle")
        k = args.check
        img, label = pool[0][0][:k], pool[0][1][:k]
        with torch.no_grad():
            _t, _o, (lin_p, clu_p), _f = model(img, img, label)
            _ot, _oo, (o_lin, o_clu), _of, _oi = oracle.wrapper_forward(state, img.cpu(), label.cpu(), cfg)
        chk = {}
        for name, got, want, hung in (("cluster", clu_p.cpu(), o_clu, True), ("linear", lin_p.cpu(), o_lin, False)):
            cg = oracle.metrics_confusion(got, label.cpu(), ncls)
            cw = oracle.metrics_confusion(want, label.cpu(), ncls)
            ig = oracle.metrics_compute(cg, hung, ncls)[0]
            iw = oracle.metrics_compute(cw, hung, ncls)[0]
            chk[name] = {"miou_hip": round(float(ig), 4), "miou_oracle": round(float(iw), 4),
                         "pixel_agreement": round(float((got == want).float().mean()), 6),
                         "confusion_l1_diff": int((cg - cw).abs().sum()), "pixels": int(got.numel())}
        line["oracle_check"] = {"images": k, **chk}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
