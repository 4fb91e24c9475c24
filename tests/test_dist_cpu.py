"""The N > 1 path on CPU: two gloo ranks exercise the collectives helpers and the image sharding."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG_NAME, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    du = importlib.import_module(PKG_NAME + ".dist_utils")
    try:
        assert du.is_distributed_set() and du.get_world_size() == world and du.get_rank() == rank
        # fused code-usage counts: one (num_pq, K) message (model/dino_pqgo.py:673 issues num_pq of them)
        counts = torch.full((8, 256), float(rank + 1))
        red = du.all_reduce_tensor(counts, op="sum")
        assert torch.equal(red, torch.full((8, 256), 3.0)) and torch.equal(counts, torch.full((8, 256), float(rank + 1)))
        assert du.all_reduce_scalar(rank + 1.0, "mean") == 1.5
        d = du.all_reduce_dict({"loss": torch.tensor([float(rank)]), "acc": 2.0 * rank, "n": rank + 1}, op="mean")
        assert torch.allclose(d["loss"], torch.tensor([0.5])) and d["acc"] == 1.0 and d["n"] in (1, 1.5)
        gathered = du.all_gather_tensor(torch.tensor([rank]))
        assert [int(t) for t in gathered] == [0, 1]
        # confusion-matrix style reduction (model/metric.py:63) and the image sharding
        conf = torch.zeros(27, 27, dtype=torch.int64)
        for i in du.shard_range(11):
            conf[i % 27, i % 27] += 1
        total = du.all_reduce_tensor(conf, op="sum")
        assert int(total.sum()) == 11
        assert sorted(list(du.shard_range(11, 0, 2)) + list(du.shard_range(11, 1, 2))) == list(range(11))
        du.barrier()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_collectives():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_helpers_are_noops(pkg):
    du = pkg.dist_utils
    t = torch.ones(3)
    assert du.all_reduce_tensor(t) is t and du.all_reduce_scalar(2.5) == 2.5
    assert du.all_reduce_dict({"a": 1.0}) == {"a": 1.0} and du.get_world_size() == 1 and du.is_master()
    assert list(du.shard_range(5)) == [0, 1, 2, 3, 4]
    assert du.set_dist("cpu") == (torch.device("cpu"), 0)
    with pytest.raises(ValueError):
        du.set_dist("tpu")
