"""The N > 1 path on CPU: world_size-2 (and 3, ragged) gloo groups.

The GPU evaluation itself cannot run here (there is no CPU fallback), so each rank evaluates
its shard with the oracle -- the checker -- and the test verifies what the multi-GPU bench relies
on: shards partition the batch, the gather reassembles rows in item order (ragged last shard
included) and the timing reduction is a max over ranks.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaast_amd.sharding import gather_rows, max_over_ranks, shard_range
from helpers import full_grades, oracle_eval_batch, rows_of


def _cfg1(B):
    a, b, c = (B.input(s, full_grades(3), 3) for s in range(3))
    return (a + b * c).g(2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)          # every rank generates the same global inputs
        rows = {s: rows_of(3, full_grades(3), batch, rng) for s in range(3)}
        lo, hi = shard_range(batch, rank, world)
        local = {s: r[lo:hi] for s, r in rows.items()}
        if hi > lo:
            out, _ = oracle_eval_batch(_cfg1, 3, local, hi - lo)
        else:
            out = np.zeros((0, 3))
        gathered = gather_rows(torch.from_numpy(np.ascontiguousarray(out)), batch, dst=0)
        t = max_over_ranks(0.5 + rank)
        if rank == 0:
            q.put((gathered.numpy(), t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 64), (2, 7), (3, 10)])
def test_sharded_eval_and_gather_matches_unsharded(world, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(1)
    rows = {s: rows_of(3, full_grades(3), batch, rng) for s in range(3)}
    want, _ = oracle_eval_batch(_cfg1, 3, rows, batch)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert t == 0.5 + (world - 1)


def test_shard_ranges_partition_the_batch():
    for batch in (0, 1, 7, 64, 65536, 1048576 + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= -(-batch // world) for lo, hi in spans)
