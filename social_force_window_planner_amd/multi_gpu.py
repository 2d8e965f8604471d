"""Multi-GPU sharding of the (v,w) grid.

The candidate trajectories are independent, so the linvel rows (outer loop of
reference src/sfw_planner.cpp:345) are split into contiguous blocks, one block
per rank = per GPU; the read-only inputs (costmap, agents) are replicated by
each rank's own H2D copy.  The only exchange step is picking the global best:
one all-reduce(min) over a [world, 4] float64 tensor in which every rank fills
its own row with its sfw_best_key and +inf elsewhere — the result is the table
of all local keys, and the lexicographic minimum of its rows is exactly the
reference's selection order (cost up, linvel down, |angvel| up, iteration index
down).  32 bytes per rank: latency-bound, xGMI bandwidth is irrelevant.

Works with backend "nccl" (= RCCL on ROCm, tensors on the GPU) and "gloo"
(CPU tensors; used by the world_size-2 tests).
"""
from __future__ import annotations

import math

import numpy as np

INF = float("inf")


def shard_rows(nv: int, rank: int, world: int):
    """Contiguous block of linvel rows for `rank`: [r*nv/W, (r+1)*nv/W)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = (rank * nv) // world
    hi = ((rank + 1) * nv) // world
    return lo, hi


def key_from_best(best: dict, linvels, angvels, nw: int, index_base: int):
    """sfw_best_key tuple for a local selection result (index local to the shard)."""
    if best["index"] < 0:
        return (INF, INF, INF, INF)
    iv, iw = divmod(int(best["index"]), nw)
    return (float(best["cost"]), -float(linvels[iv]), abs(float(angvels[iw])),
            -float(index_base + int(best["index"])))


def lexicographic_min(rows):
    """Row index and row of the lexicographic minimum over finite-cost rows."""
    best_r, best_k = None, None
    for r, k in enumerate(rows):
        k = tuple(float(x) for x in k)
        if not math.isfinite(k[0]):
            continue
        if best_k is None or k < best_k:
            best_r, best_k = r, k
    return best_r, best_k


def exchange_best(local_key, dist, rank: int, world: int, device=None):
    """One all-reduce(min): returns (winner_rank, winner_key, table[world,4]).
    `dist` is torch.distributed (already initialised)."""
    import torch

    t = torch.full((world, 4), INF, dtype=torch.float64, device=device)
    t[rank] = torch.tensor(local_key, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    table = t.cpu().numpy()
    win_rank, win_key = lexicographic_min(table)
    return win_rank, win_key, table


class KeyExchange:
    """exchange_best with its tensors allocated once (a bench step is ~10 ms, the exchange should stay in the
    tens of microseconds): the [world,4] table on `device`, a host staging row, one all-reduce(min) per call."""

    def __init__(self, dist, rank: int, world: int, device=None):
        import torch

        self.dist, self.rank, self.world = dist, rank, world
        self.table = torch.full((world, 4), INF, dtype=torch.float64, device=device)
        on_gpu = self.table.is_cuda
        self.row = torch.empty(4, dtype=torch.float64, pin_memory=on_gpu)
        self.host = torch.empty((world, 4), dtype=torch.float64, pin_memory=on_gpu)

    def __call__(self, local_key):
        self.row[0], self.row[1], self.row[2], self.row[3] = (float(v) for v in local_key)
        self.table.fill_(INF)
        self.table[self.rank].copy_(self.row, non_blocking=True)
        self.dist.all_reduce(self.table, op=self.dist.ReduceOp.MIN)
        self.host.copy_(self.table)  # synchronises
        table = self.host.numpy()
        win_rank, win_key = lexicographic_min(table)
        return win_rank, win_key, table


def cmd_from_key(key, nw_total: int, linvels_all, angvels):
    """Global (vx, vtheta, index) from a winning key; (0, 0, -1) when none."""
    if key is None:
        return 0.0, 0.0, -1
    idx = int(round(-key[3]))
    iv, iw = divmod(idx, nw_total)
    return float(np.asarray(linvels_all)[iv]), float(np.asarray(angvels)[iw]), idx
