"""The N > 1 path on CPU: world_size-2 (and 3, ragged) gloo groups started by the SAME launcher bench.py uses
(gaast_amd.launch.spawn_ranks: one process per rank, torch.distributed environment, nobody re-execs).

The GPU evaluation itself cannot run here (there is no CPU fallback), so each rank evaluates its shard with the
oracle -- the checker -- and the test verifies what the multi-GPU bench relies on: shards partition the batch, the
chunk schedule of gaast_hip_eval_gather reassembles rows in item order (ragged shards and empty chunks included), the
timing reduction is a max over ranks, and a failing rank fails the job.  The product's own N > 1 control flow runs on the
GPU box: tests/test_gpu_bench_contract.py::test_bench_self_launches_two_ranks_and_gathers.
"""
import os
import sys

import numpy as np
import pytest

from gaast_amd.launch import launched_by_a_launcher, rank_env, spawn_ranks
from gaast_amd.sharding import chunk_span, shard_range
from helpers import full_grades, oracle_eval_batch, rows_of

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "sharding_worker.py")


def _cfg1(B):
    a, b, c = (B.input(s, full_grades(3), 3) for s in range(3))
    return (a + b * c).g(2)


@pytest.mark.parametrize("world,batch,chunks", [(2, 64, 4), (2, 7, 4), (3, 10, 3)])
def test_sharded_eval_and_chunked_gather_match_unsharded(world, batch, chunks, tmp_path):
    out = str(tmp_path / "gathered.npz")
    rc = spawn_ranks([sys.executable, WORKER, str(batch), str(chunks), out], world, timeout=300)
    assert rc == 0
    got = np.load(out)
    rng = np.random.default_rng(1)
    rows = {s: rows_of(3, full_grades(3), batch, rng) for s in range(3)}
    want, _ = oracle_eval_batch(_cfg1, 3, rows, batch)
    assert got["whole"].shape == want.shape and np.array_equal(got["whole"], want)
    assert np.array_equal(got["chunked"], want)
    assert float(got["t"]) == 0.5 + (world - 1)
    assert int(got["counts"].sum()) == batch


def test_a_failing_rank_fails_the_job_and_ends_the_others(tmp_path):
    rc = spawn_ranks([sys.executable, WORKER, "8", "2", str(tmp_path / "x.npz"), "1"], 2, timeout=120)
    assert rc == 3
    assert not os.path.exists(tmp_path / "x.npz")


def test_rank_environment():
    env = rank_env(1, 4, 2345, base={})
    assert env["RANK"] == "1" and env["LOCAL_RANK"] == "1" and env["WORLD_SIZE"] == "4"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "2345" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not launched_by_a_launcher(2) or os.environ.get("WORLD_SIZE") == "2"


def test_shard_ranges_partition_the_batch():
    for batch in (0, 1, 7, 64, 65536, 1048576 + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= -(-batch // world) for lo, hi in spans)


def test_chunk_spans_partition_a_shard():
    for count in (0, 1, 5, 131072, 131073):
        for chunks in (1, 3, 4, 8):
            spans = [chunk_span(count, chunks, c) for c in range(chunks)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
