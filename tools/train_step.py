"""PQ training step of the reference's harness (train.py:74-98: forward -> backward -> clip -> Adam
step) on synthetic data, one process per GPU; under torchrun the wrapper is wrapped in
DistributedDataParallel exactly as train.py:415-417 does and gradients are all-reduced over RCCL.
Prints one JSON line from rank 0 (images/s over all ranks, ms per step).

  python tools/train_step.py --batch 16 --res 224 --steps 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_step.py
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
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
    pkg = importlib.import_module(PKG)
    cfg = pkg.configs.pqgo_config(need_initialized="uni")
    wrapper = pkg.build_model(cfg)
    wrapper.load_state_dict(pkg.synth.synth_pqgo_state(cfg, seed=1234), strict=True)
    wrapper = wrapper.to(dev).train()
    model = wrapper
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(wrapper, device_ids=[local_rank])
    params = [p for p in wrapper.parameters() if p.requires_grad]
    optim = torch.optim.Adam(params, lr=3e-4)   # config/pqgo_baseline.yaml:154-165
    img = pkg.synth.synth_images(args.batch, args.res, args.res, seed=1 + rank).to(dev)
    pos = pkg.synth.synth_images(args.batch, args.res, args.res, seed=101 + rank).to(dev)
    label = pkg.synth.synth_labels(args.batch, args.res, args.res, cfg["num_classes"], seed=1 + rank).to(dev)

    def step(it):
        optim.zero_grad(set_to_none=True)
        total, out, _preds, _f = model(img=img, aug_img=img, label=label, img_pos=pos, it=it)
        total.backward()
        torch.nn.utils.clip_grad_norm_(wrapper.model.parameters(), max_norm=10.0)   # train.py:93, clip_grad 10
        optim.step()
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"workload": f"PQ training step, ViT-S/8 frozen + head/codebooks/probes trained, "
                                      f"{args.batch} x {args.res}^2 per GPU (+ positive pair), synthetic",
                          "n_gpus": world, "images_per_s": round(world * args.batch * args.steps / dt, 1),
                          "ms_per_step": round(dt / args.steps * 1e3, 2),
                          "losses": {k: round(float(torch.as_tensor(v).detach()), 5) for k, v in out.items() if k != "end_time"}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
