"""CPU: the N>1 path with gloo, world_size 2 — shards equal the rows of the single-process batch and
the final gather restores scenario order bit for bit.  (The arithmetic here is the CPU oracle's: this
test is about sharding and ordering, the GPU arithmetic is covered by the gpu-marked tests.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import host_scenarios, params_for
from mpc_amd.sharding import gather_rows, shard_range


def test_shard_range_covers_everything():
    for total, world in [(10, 2), (11, 2), (7, 8), (65536, 8), (1, 3)]:
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, H, queue):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mpar = params_for(H)
        first, count = shard_range(total, world, rank)
        co, hs = host_scenarios(mpar, count, motion="angular", cfg_id=5, first=first)
        cost, grad, con, _ = co.eval_fd(hs.as_oracle_dict(), want_jac=False)
        local = torch.as_tensor(np.concatenate([cost[:, None], grad[:, :6], con[:, :3]], axis=1))
        full = gather_rows(local, total)
        if rank == 0:
            queue.put(full.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [12, 13])
def test_two_rank_shards_equal_single_process(total):
    H, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mpar = params_for(H)
    co, hs = host_scenarios(mpar, total, motion="angular", cfg_id=5)
    cost, grad, con, _ = co.eval_fd(hs.as_oracle_dict(), want_jac=False)
    want = np.concatenate([cost[:, None], grad[:, :6], con[:, :3]], axis=1)
    np.testing.assert_array_equal(got, want)
