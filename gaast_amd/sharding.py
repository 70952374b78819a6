"""Batch sharding across the GPUs of one node: one process per GPU, contiguous item ranges.

Every batch item is an independent evaluation of the same SpecializedAst (the reference keeps
no cross-item state: src/eval.rs:16 creates a fresh cache per call), so the data path needs no
collective.  The only exchange is the final gather of the per-rank result rows, done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
from __future__ import annotations


def shard_range(batch: int, rank: int, world: int):
    """Contiguous [start, stop) of the items rank `rank` evaluates: ceil(batch / world) each."""
    per = -(-batch // world)
    start = min(rank * per, batch)
    return start, min(start + per, batch)


def chunk_span(count: int, n_chunks: int, c: int):
    """[lo, hi) of chunk c when `count` items are cut into n_chunks contiguous chunks (the schedule of
    gaast_hip_eval_gather, runtime.hip: chunk_span)."""
    per = -(-count // n_chunks) if n_chunks else count
    lo = min(c * per, count)
    return lo, min(lo + per, count)


def gather_rows(local_rows, batch: int, dst: int = 0, group=None):
    """Gather the per-rank result rows ([n_local, row_len] tensors) to `dst`, in item order.

    Ranks may hold different numbers of rows (ragged last shard): rows are padded to the
    common shard size for the collective and trimmed afterwards.  Returns the [batch, row_len]
    tensor on `dst`, None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-batch // world)
    row_len = local_rows.shape[1]
    send = local_rows
    if send.shape[0] != per:
        pad = torch.zeros((per - send.shape[0], row_len), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.cat(bufs, dim=0)
    return out[:batch]


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The timing every rank reports: the slowest rank's."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
