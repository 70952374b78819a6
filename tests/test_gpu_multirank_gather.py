"""The library's OWN multi-rank gather, executed with world > 1 (BASELINE configs[3] is the one configuration whose
exchange cannot run on a one-GPU box through RCCL: RCCL refuses two ranks on one device).

Every rank is a fresh process of the plain C host tests/cpp/gather_host.c, started by gaast_amd.launch.spawn_ranks;
the nccl* entry points come from the test transport tests/cpp/rccl_stub.c through gaast_hip_comm_set_library, so
gaast_hip_comm_init / gaast_hip_comm_count_ranks / gaast_hip_eval_gather / gaast_hip_gather_rows run exactly the code
an 8-GPU node runs: chunk_span schedule, ncclSend on the peers, grouped ncclRecv on the root, chunk events, the join of
the communicator's stream.  The gathered rows must equal the oracle's rows for the whole batch, bit for bit, and the
rows a single process computes with gaast_hip_eval (tests/cpp/abi_host.c)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import build_c_host, build_rccl_stub, oracle_eval_batch, rows_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_transport_and_hosts_compile(tmp_path):
    """CPU part: the stub exports the nine entry points comm.cpp resolves; the C hosts link."""
    so = build_rccl_stub(tmp_path)
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclAllReduce",
                 "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert f" T {name}" in syms, name
    exe = build_c_host("gather_host", tmp_path)
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 2 and "usage" in run.stderr


def test_set_library_is_checked_without_a_gpu():
    """gaast_hip_comm_set_library: a missing file is reported when the communicator is first used, not silently replaced
    by librccl; no GPU needed to reach the loader."""
    import gaast_amd as ga
    L = ga.lib()
    assert L.gaast_hip_comm_set_library(b"/nonexistent/librccl_stub.so") == 0
    try:
        assert L.gaast_hip_comm_set_library(None) == 0     # back to the default: nothing was loaded yet
    finally:
        L.gaast_hip_comm_set_library(None)


def _run_ranks(tmp_path, counts, root, n_chunks, alias, seed):
    from gaast_amd.launch import spawn_ranks
    world, total = len(counts), sum(counts)
    so = build_rccl_stub(tmp_path)
    exe = build_c_host("gather_host", tmp_path)
    rng = np.random.default_rng(seed)
    rows = {0: rows_of(5, [0, 2, 4], total, rng), 1: rows_of(5, [1], total, rng)}
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(rows[0].tobytes())
        f.write(rows[1].tobytes())
    out = tmp_path / "out.bin"
    rc = spawn_ranks([exe, str(tmp_path / "in.bin"), str(out), str(tmp_path / "comm.id"), so, str(root), str(n_chunks),
                      str(int(alias))] + [str(c) for c in counts], world, timeout=420)
    assert rc == 0, f"a rank failed (exit code {rc})"
    got = np.fromfile(out, dtype=np.float64).reshape(total, 16)
    build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, [0, 2, 4], 5), B.input(1, [1], 5))
    want, mask = oracle_eval_batch(build, [1.0, 1.0, 1.0, 1.0, -1.0], rows, total)
    assert mask == 0x2A
    assert np.array_equal(got, want), "gathered rows differ from the oracle's"
    return rows, got


@pytest.mark.gpu
@pytest.mark.parametrize("counts,root,n_chunks,alias", [
    ([300, 300], 0, 4, True),        # the bench's shape: equal shards, root's rows live inside `gathered`
    ([4, 3], 0, 4, False),           # ragged: 7 items, 4 chunks -> chunks of 1 (and an empty one on rank 1)
    ([5, 2, 7], 1, 3, True),         # world 3, a root other than 0, aliasing its middle range
    ([1, 0, 6], 2, 2, False),        # a rank with nothing to send
    ([1000, 999, 37], 0, 1, False),  # one chunk = the blocking schedule through the overlapped entry point
])
def test_library_gather_with_several_ranks_is_bit_exact(tmp_path, counts, root, n_chunks, alias):
    _run_ranks(tmp_path, counts, root, n_chunks, alias, seed=100 + len(counts) + root)


@pytest.mark.gpu
def test_gathered_rows_equal_a_single_process_eval(tmp_path):
    """... and bit-equal to gaast_hip_eval of the whole batch in ONE process (tests/cpp/abi_host.c)."""
    rows, got = _run_ranks(tmp_path, [129, 128], 0, 4, True, seed=7)
    exe = build_c_host("abi_host", tmp_path)
    one = tmp_path / "single.bin"
    run = subprocess.run([exe, str(tmp_path / "in.bin"), str(one), str(257)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert np.array_equal(np.fromfile(one, dtype=np.float64).reshape(257, 16), got)


@pytest.mark.gpu
def test_a_gather_whose_sends_and_receives_do_not_pair_up_fails_loudly(tmp_path):
    """The positive tests mean something only if a mis-pairing is noticed: the root under-counts one peer's rows, its
    receive meets a send of another size, the transport reports it and the job ends non-zero (no hang)."""
    from gaast_amd.launch import spawn_ranks
    so = build_rccl_stub(tmp_path)
    exe = build_c_host("gather_host", tmp_path)
    rng = np.random.default_rng(3)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(rows_of(5, [0, 2, 4], 20, rng).tobytes())
        f.write(rows_of(5, [1], 20, rng).tobytes())
    rc = spawn_ranks([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "comm.id"), so, "0", "1", "0",
                      "10", "10"], 2, env_extra={"GAAST_TEST_ROOT_MISCOUNT": "1", "GAAST_RCCL_STUB_TIMEOUT_S": "20"}, timeout=300)
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,root,failing", [(2, 0, 1), (3, 1, 2)])
def test_one_rank_failing_its_evaluation_is_reported_by_every_rank(tmp_path, world, root, failing):
    """A failure of gaast_hip_eval_gather is COLLECTIVE (include/gaast_hip.h): a non-root rank's evaluation fails
    (GAAST_FLAG_DEBUG_FAIL_EVAL on that rank only); it still posts its transfers, nobody hangs, and the root -- which
    now holds stale rows of that rank -- returns non-zero like every other rank instead of GAAST_OK."""
    from gaast_amd.launch import free_port, rank_env
    so = build_rccl_stub(tmp_path)
    exe = build_c_host("gather_host", tmp_path)
    counts = [9, 5, 7][:world]
    rng = np.random.default_rng(11)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(rows_of(5, [0, 2, 4], sum(counts), rng).tobytes())
        f.write(rows_of(5, [1], sum(counts), rng).tobytes())
    port = free_port()
    procs = []
    for rank in range(world):
        env = rank_env(rank, world, port)
        env.update({"GAAST_TEST_FAIL_EVAL_RANK": str(failing), "GAAST_RCCL_STUB_TIMEOUT_S": "60"})
        procs.append(subprocess.Popen([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(tmp_path / "comm.id"), so,
                                       str(root), "3", "0"] + [str(c) for c in counts], env=env, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    codes = [p.returncode for p in procs]
    # 42 = "gaast_hip_eval_gather returned the status the header promises": GAAST_ERR_HIP on the failing rank, GAAST_ERR_RCCL elsewhere
    assert codes == [42] * world, (codes, [o[1][-500:] for o in outs])
    assert "injected evaluation failure" in outs[failing][1] and "another rank failed" in outs[root][1]
