"""Entry sharding across ranks and the ordered gather of compressed shards (one process per GPU).

The reference is data-parallel one-entry-per-task on CPU threads and re-orders results by index on one thread
(cli/src/command/core.rs:496-537 spawn_entry_results, :471-493 drain_entry_results, core/iter.rs ReorderByIndex).
Here rank r owns a CONTIGUOUS index range balanced by input bytes, so the final stream is the concatenation of the
ranks' outputs in rank order -- the only exchange is that gather (RCCL send/recv over xGMI on GPUs, gloo in tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def partition_entries(sizes: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) per rank, balanced by bytes (greedy on the running prefix)."""
    n, total = len(sizes), sum(sizes)
    bounds, acc, start = [], 0, 0
    for r in range(world):
        target = total * (r + 1) / world
        end = start
        while end < n and (acc + sizes[end] <= target or r == world - 1):
            acc += sizes[end]
            end += 1
        if r < world - 1 and end < n and end == start and n - start > world - 1 - r:
            acc += sizes[end]
            end += 1
        bounds.append((start, end))
        start = end
    bounds[-1] = (bounds[-1][0], n)
    return bounds


def piece_ranges(rank: int, world: int, pieces: int, per_piece: int) -> List[Tuple[int, int]]:
    """Weak-scaling layout with piecewise gathers: piece h of rank r holds the entries [(h * world + r) * per_piece, ... + per_piece).
    In archive order all ranks' pieces 0 come first (in rank order), then all pieces 1, ...: each piece can be gathered in rank order as
    soon as it is compressed, and the concatenation of the gathered pieces is the archive."""
    return [((h * world + rank) * per_piece, (h * world + rank + 1) * per_piece) for h in range(pieces)]


class GatherOverflow(RuntimeError):
    """The parts do not fit the destination rank 0 offered (and growing it was not allowed).  Raised on EVERY rank -- the capacity travels with the sizes --
    before anything is sent, so the job sees an error, never a hang (pna_gpu_gather_ordered's PNA_E_DSTSIZE; include/pna_gpu.h)."""

    def __init__(self, need: int, cap: int, sizes):
        super().__init__(f"ordered gather: {need} bytes do not fit the {cap} offered on rank 0")
        self.need, self.cap, self.sizes = need, cap, sizes


def _exchange_sizes(local, local_bytes: int, rank: int, world: int, out, grow: bool):
    """All-gather of (part size, capacity offered): every rank gets the same pairs, hence the same verdict.  Capacity -1 = rank 0 may allocate."""
    import torch
    import torch.distributed as dist
    cap = -1 if (grow or rank != 0) else (out.numel() if out is not None else 0)
    pairs = torch.zeros(2 * world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(pairs, torch.tensor([local_bytes, cap], dtype=torch.int64, device=local.device))
    pl = [int(x) for x in pairs.tolist()]
    sizes, cap0 = pl[0::2], pl[1]
    need = sum(sizes)
    if cap0 >= 0 and need > cap0:
        raise GatherOverflow(need, cap0, sizes)
    return sizes, need


def gather_ordered(local, local_bytes: int, rank: int, world: int, out=None, grow: bool = True):
    """Gather every rank's first `local_bytes` bytes of the uint8 tensor `local` onto rank 0, in rank order.

    Returns (gathered bytes, sizes) on rank 0 and (None, sizes) elsewhere.  Works for CUDA tensors over the nccl (=RCCL)
    backend and CPU tensors over gloo.  The synchronous form of gather_ordered_start / gather_ordered_wait."""
    return gather_ordered_wait(gather_ordered_start(local, local_bytes, rank, world, out=out, grow=grow))[::2]


def gather_ordered_start(local, local_bytes: int, rank: int, world: int, out=None, grow: bool = True):
    """Asynchronous form of gather_ordered: posts the size exchange and the sends / receives and returns a handle for
    gather_ordered_wait().  `local` (and `out` on rank 0) must stay untouched until the wait returns -- callers that keep
    producing double-buffer `local`, so the gather of shard k overlaps the compression of shard k+1.
    `out` (rank 0): a destination to reuse; when it is too small a larger one is allocated if `grow`, else GatherOverflow is raised on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1:
        if out is not None and not grow and out.numel() < local_bytes:
            raise GatherOverflow(local_bytes, out.numel(), [local_bytes])
        return {"works": [], "out": local[:local_bytes], "buf": local, "sizes": [local_bytes]}
    sizes, need = _exchange_sizes(local, local_bytes, rank, world, out, grow)
    if rank == 0:
        if out is None or out.numel() < need:
            out = torch.empty(need, dtype=torch.uint8, device=local.device)
        out[:sizes[0]].copy_(local[:sizes[0]], non_blocking=True)
        ops, pos = [], sizes[0]
        for r in range(1, world):
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, out[pos:pos + sizes[r]], r))
            pos += sizes[r]
        return {"works": dist.batch_isend_irecv(ops) if ops else [], "out": out[:need], "buf": out, "sizes": sizes}
    works = dist.batch_isend_irecv([dist.P2POp(dist.isend, local[:local_bytes], 0)]) if local_bytes else []
    return {"works": works, "out": None, "buf": None, "sizes": sizes}


def gather_ordered_wait(handle):
    """Completes a gather_ordered_start(); returns (gathered, buffer, sizes): `gathered` = exactly the gathered bytes (rank 0; None elsewhere) -- the same
    on every path --, `buffer` = the tensor behind it, to offer as `out` next time (not the caller's own when that was too small)."""
    for w in handle["works"]:
        w.wait()
    return handle["out"], handle["buf"], handle["sizes"]
