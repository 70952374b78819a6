"""One rank of the CPU rehearsal of the multi-GPU path (started by gaast_amd.launch.spawn_ranks, like bench.py's ranks).

    python sharding_worker.py <batch> <n_chunks> <out.npz> [fail_rank]

The GPU evaluation cannot run here (there is no CPU fallback in the product), so the rank evaluates its shard with the
oracle -- the checker -- and then does what the bench does around the evaluation: contiguous shards, the chunk schedule
of gaast_hip_eval_gather, a gather of every chunk to rank 0 in item order, a max-over-ranks timing reduction.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gaast_amd.sharding import chunk_span, gather_rows, max_over_ranks, shard_range  # noqa: E402
from helpers import full_grades, oracle_eval_batch, rows_of  # noqa: E402


def cfg1(B):
    a, b, c = (B.input(s, full_grades(3), 3) for s in range(3))
    return (a + b * c).g(2)


def main():
    batch, n_chunks, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    fail_rank = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if rank == fail_rank:
        sys.exit(3)                       # a rank that dies before the rendezvous: the launcher must end the others
    dist.init_process_group("gloo")       # MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE from the launcher
    try:
        rng = np.random.default_rng(1)    # every rank generates the same global inputs
        rows = {s: rows_of(3, full_grades(3), batch, rng) for s in range(3)}
        lo, hi = shard_range(batch, rank, world)
        counts = [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
        local = np.zeros((hi - lo, 3))
        pieces = []
        for c in range(n_chunks):         # chunk c is "evaluated", then gathered while the next one would compute
            clo, chi = chunk_span(hi - lo, n_chunks, c)
            if chi > clo:
                local[clo:chi], _ = oracle_eval_batch(cfg1, 3, {s: r[lo + clo:lo + chi] for s, r in rows.items()}, chi - clo)
            # every rank contributes its chunk c: rank r's rows land at offset(r) + chunk offset on the root
            sizes = [chunk_span(n, n_chunks, c) for n in counts]
            per = max(b - a for a, b in sizes)
            send = torch.zeros((per, 3), dtype=torch.float64)
            send[:chi - clo] = torch.from_numpy(local[clo:chi])
            bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
            dist.gather(send, bufs, dst=0)
            if rank == 0:
                pieces.append([(sizes[r], bufs[r]) for r in range(world)])
        whole = gather_rows(torch.from_numpy(np.ascontiguousarray(local)), batch, dst=0)
        t = max_over_ranks(0.5 + rank)
        if rank == 0:
            chunked = np.zeros((batch, 3))
            first = 0
            for r in range(world):
                for c in range(n_chunks):
                    (a, b), buf = pieces[c][r]
                    chunked[first + a:first + b] = buf[:b - a].numpy()
                first += counts[r]
            np.savez(out_path, whole=whole.numpy(), chunked=chunked, t=t, counts=np.array(counts))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
