"""Node partitioning over the GPUs of one node (one process per GPU, RCCL/xGMI).

The reference is single-device (``ex_acm3025.py:11``); this is new design
(SURVEY.md section 8e).  Rows (destination nodes) are split into contiguous
blocks of ``shard = ceil(N / world)`` rows, so a node's global id is also its
row in any all-gathered table of ``world * shard`` rows -- ``colidx`` needs no
remapping.  Exchange steps of one training step:

  forward   all-gather of the projected rows H per meta-path (the dropout keep
            bits travel inside the rows; f2 is recomputed by the consumer)
  backward  all-gather of [g | stats] per meta-path (the transposed-graph pass
            gathers from every destination), then ONE all-reduce of the flat
            parameter-gradient buffer.

The semantic attention, classifier and loss are row-local (the code's softmax
over meta-paths is per node), so they need no collective.  On a uniformly
random graph every remote row is needed by someone, so the halo exchange
degenerates to an all-gather; that is what is implemented.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .graph import CSRGraph


class NodePartition:
    def __init__(self, n_global: int, rank: int | None = None, world: int | None = None,
                 group=None):
        if rank is None or world is None:
            if dist.is_available() and dist.is_initialized():
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world = 0, 1
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.n_global = int(n_global)
        self.rank, self.world, self.group = int(rank), int(world), group
        self.shard = (self.n_global + self.world - 1) // self.world
        self.row_start = min(self.rank * self.shard, self.n_global)
        self.row_end = min(self.row_start + self.shard, self.n_global)
        self.n_local = self.row_end - self.row_start
        self.n_table = self.shard * self.world     # rows of an all-gathered table
        # HAN_FORCE_COLLECTIVES=1 runs the collectives even on a 1-rank group (used to
        # exercise the RCCL calls on a single-GPU box)
        self.active = self.world > 1 or os.environ.get("HAN_FORCE_COLLECTIVES") == "1"

    # ---- graph sharding -------------------------------------------------------
    def shard_graph(self, g: CSRGraph) -> tuple[CSRGraph, CSRGraph]:
        """Global CSR (n_global x n_global) -> (rows_local, cols_local):
        rows_local: CSR of this rank's destination rows, colidx = global ids;
        cols_local: transposed graph restricted to this rank's source rows,
        its colidx = global destination ids."""
        if g.n_rows != self.n_global or g.n_cols != self.n_global:
            raise ValueError("shard_graph expects the global square graph")
        rows_local = _row_block(g, self.row_start, self.row_end, self.n_table)
        cols_local = _row_block(g.transpose(), self.row_start, self.row_end, self.n_table)
        return rows_local, cols_local

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.row_start:self.row_end]

    # ---- collectives ------------------------------------------------------------
    def _backend(self) -> str:
        return dist.get_backend(self.group)

    def all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """local (n_local, ...) -> table (world*shard, ...); rows past n_global
        (padding of the last shard) hold zeros and are never indexed."""
        return self.all_gather_rows_async(local).wait()

    def all_gather_rows_async(self, local: torch.Tensor) -> "GatheredTable":
        """Start the all-gather and return a handle; `.wait()` yields the table.
        Under RCCL the collective runs on the communicator's stream (after the
        producer kernels already queued on the current stream) and overlaps with
        whatever is launched before `.wait()`; under gloo it completes here."""
        if not self.active:
            return GatheredTable(local, None)
        tail = tuple(local.shape[1:])
        if local.shape[0] != self.shard:
            padded = local.new_zeros((self.shard,) + tail)
            padded[:local.shape[0]] = local
        else:
            padded = local.contiguous()
        table = local.new_empty((self.n_table,) + tail)
        if self._backend() == "nccl":
            work = dist.all_gather_into_tensor(table, padded, group=self.group, async_op=True)
            return GatheredTable(table, work, keep=padded)
        # gloo (CPU rehearsal / single-GPU multi-process tests): stage through host
        src = padded.cpu()
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self.group)
        table.copy_(torch.cat(parts, 0))
        return GatheredTable(table, None)

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        if not self.active:
            return flat
        if self._backend() == "nccl" or not flat.is_cuda:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        else:
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host)
        return flat


class GatheredTable:
    """Handle of an (possibly in-flight) all-gathered table."""

    def __init__(self, table, work, keep=None):
        self.table, self.work, self.keep = table, work, keep

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()          # the current stream waits for the collective
            self.work, self.keep = None, None
        return self.table


def _row_block(g: CSRGraph, r0: int, r1: int, n_cols: int) -> CSRGraph:
    rp = g.rowptr[r0:r1 + 1]
    s, e = int(rp[0]), int(rp[-1])
    return CSRGraph((rp - rp[0]).contiguous(), g.colidx[s:e].contiguous(), n_cols, validate=False)
