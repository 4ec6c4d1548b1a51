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
over meta-paths is per node), so they need no collective.

Two exchange modes per (meta-path, direction), chosen at setup by `plan_exchange`:
  * halo  -- only the remote rows this rank's edges actually reference travel:
            precomputed send lists -> pack -> all-to-all-v -> a [local | halo]
            table read through a remapped colidx (`HaloPlan`);
  * all-gather -- when the halo is most of the remote rows anyway (a uniformly
            random graph at deg 50 references 99.8 % of them) the pack and the
            remap buy nothing, so the whole shard is all-gathered.
"""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

from .graph import CSRGraph


class CommStats:
    """What a measured region exchanged (bench.py's N > 1 line): bytes this rank RECEIVED in table
    exchanges (all-gather: the other ranks' blocks; halo: the halo rows), the size of the all-reduced
    gradient buffer, and HIP events on the compute stream around every wait for a collective -- the time
    the compute stream stood still for the links (0 when the exchange was fully hidden behind kernels).
    Attach with `part.comm = CommStats()`; None (the default) records nothing."""

    def __init__(self):
        self.bytes_received = 0
        self.allreduce_bytes = 0
        self.exchanges = 0
        self.host_ms = 0.0          # host-staged (gloo rehearsal) collectives complete on the host: wall time there
        self._events = []

    def received(self, nbytes: int):
        self.bytes_received += int(nbytes)
        self.exchanges += 1

    def bracket(self, device):
        """(start, stop) events on the current stream of `device`, or None off the GPU."""
        if device.type != "cuda":
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._events.append((e0, e1))
        return e0, e1

    def wait_ms(self) -> float:
        """Sum of the bracketed waits (call after a device synchronize)."""
        return float(sum(e0.elapsed_time(e1) for e0, e1 in self._events)) + self.host_ms


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
        self._bufs: dict = {}       # exchange tables, allocated once per (tag, shape, dtype) and reused every step
        self.comm: CommStats | None = None

    def buffer(self, tag, shape, dtype, device) -> torch.Tensor:
        """Persistent exchange buffer.  `tag` names the use (direction, layer, meta-path): a buffer is
        rewritten by the next exchange with the same tag, after its consumer kernel was queued."""
        if tag is None:
            return torch.empty(shape, dtype=dtype, device=device)
        key = (tag, tuple(shape), dtype, str(device))
        t = self._bufs.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    def drop_buffers(self, tag) -> None:
        """Release the persistent buffers registered under `tag`."""
        self._bufs = {k: v for k, v in self._bufs.items() if k[0] != tag and k[0] != (tag, "pad")}

    # ---- graph sharding -------------------------------------------------------
    def shard_graph(self, g: CSRGraph) -> tuple[CSRGraph, CSRGraph]:
        """Global CSR (n_global x n_global) -> (rows_local, cols_local), see shard_local_graph.
        Convenience for callers that hold the whole graph (small data sets, tests): only this
        rank's row block is touched; at scale every rank builds or loads its own rows and calls
        shard_local_graph directly, so that no rank ever holds the global graph."""
        if g.n_rows != self.n_global or g.n_cols != self.n_global:
            raise ValueError("shard_graph expects the global square graph")
        rows_local = _row_block(g, self.row_start, self.row_end, self.n_table)
        # the transposed shard straight from the global graph, no collective: the edges whose source
        # column is local, their destination rows recovered from rowptr by binary search, one
        # stable sort of E/G elements
        dev = g.device
        col = g.colidx
        sel = torch.nonzero((col >= self.row_start) & (col < self.row_end)).flatten()
        dst = (torch.searchsorted(g.rowptr, sel, right=True) - 1).to(torch.int32)
        src = (col[sel] - self.row_start).long()
        order = torch.sort(src, stable=True).indices          # destinations stay ascending per source
        rowidx = dst[order].contiguous()
        counts = torch.bincount(src, minlength=self.n_local)
        colptr = torch.zeros(self.n_local + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=colptr[1:])
        vals = g.values[sel][order].contiguous() if g.values is not None else None
        return rows_local, CSRGraph(colptr, rowidx, self.n_table, validate=False, values=vals, row_base=self.row_start)

    def shard_local_graph(self, rows_local: CSRGraph) -> tuple[CSRGraph, CSRGraph]:
        """rows_local: CSR of THIS rank's destination rows [row_start, row_end) with GLOBAL column ids.
        Returns (rows_local, cols_local):
        rows_local: the same rows as a table-indexing graph (n_cols = world * shard);
        cols_local: the transposed graph restricted to this rank's source rows -- for every local
        source j the global destination ids i in ascending order -- built from an all-to-all-v of
        the edges (each edge (i, j) travels once, to the owner of j) and an E/G-element stable sort.
        Collective: every rank must call it for the same meta-path in the same order."""
        if rows_local.n_rows != self.n_local:
            raise ValueError(f"expected this rank's {self.n_local} rows, got {rows_local.n_rows}")
        dev = rows_local.device
        rows_local = CSRGraph(rows_local.rowptr, rows_local.colidx, self.n_table, validate=False,
                              values=rows_local.values, row_base=self.row_start)
        col = rows_local.colidx
        if col.numel() and (int(col.min()) < 0 or int(col.max()) >= self.n_global):
            raise ValueError("rows_local.colidx must hold global node ids")
        dst = torch.repeat_interleave(torch.arange(self.row_start, self.row_end, device=dev, dtype=torch.int32),
                                      rows_local.degrees())
        owner = torch.div(col, self.shard, rounding_mode="floor").long()
        order = torch.sort(owner, stable=True).indices          # grouped by owner, local edge order kept
        send_counts = torch.bincount(owner, minlength=self.world)
        del owner
        payload = torch.stack([dst[order], col[order]], dim=1).contiguous()     # (E_local, 2) int32
        del dst
        recv_counts = self.all_to_all_v(send_counts.new_empty(self.world), send_counts, [1] * self.world,
                                        [1] * self.world)
        send_splits = [int(c) for c in send_counts.tolist()]
        recv_splits = [int(c) for c in recv_counts.tolist()]
        edges = payload.new_empty((sum(recv_splits), 2))
        self.all_to_all_v(edges, payload, recv_splits, send_splits)
        del payload
        vals = None
        if rows_local.values is not None:
            vals = rows_local.values.new_empty(sum(recv_splits))
            self.all_to_all_v(vals, rows_local.values[order].contiguous(), recv_splits, send_splits)
        del order
        # arrivals are ordered by sender rank (ascending row blocks) and, within a sender, by its CSR
        # order (ascending destination): a stable sort by source leaves each source's destinations ascending
        src = (edges[:, 1] - self.row_start).long()
        order2 = torch.sort(src, stable=True).indices
        rowidx = edges[:, 0][order2].contiguous()
        counts = torch.bincount(src, minlength=self.n_local)
        colptr = torch.zeros(self.n_local + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=colptr[1:])
        cols_local = CSRGraph(colptr, rowidx, self.n_table, validate=False,
                              values=vals[order2].contiguous() if vals is not None else None, row_base=self.row_start)
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

    def all_gather_rows_async(self, local: torch.Tensor, tag=None, rows_per_rank: int | None = None) -> "GatheredTable":
        """Start the all-gather and return a handle; `.wait()` yields the table.
        Under RCCL the collective runs on the communicator's stream (after the
        producer kernels already queued on the current stream) and overlaps with
        whatever is launched before `.wait()`; under gloo it completes here.
        tag: reuse the persistent table of that name instead of allocating one per step.
        rows_per_rank: block height every rank contributes (default: the shard); `local` is zero-padded to it."""
        if not self.active:
            return GatheredTable(local, None)
        per = self.shard if rows_per_rank is None else int(rows_per_rank)
        tail = tuple(local.shape[1:])
        if local.shape[0] != per:
            padded = self.buffer(None if tag is None else (tag, "pad"), (per,) + tail, local.dtype, local.device)
            padded[:local.shape[0]] = local
            padded[local.shape[0]:] = 0
        else:
            padded = local.contiguous()
        table = self.buffer(tag, (per * self.world,) + tail, local.dtype, local.device)
        if self.comm is not None:
            self.comm.received((self.world - 1) * padded.numel() * padded.element_size())
        if self._backend() == "nccl":
            work = dist.all_gather_into_tensor(table, padded, group=self.group, async_op=True)
            return GatheredTable(table, work, keep=padded, comm=self.comm)
        # gloo (CPU rehearsal / single-GPU multi-process tests): stage through host, as raw bytes
        t0 = time.perf_counter()
        src = padded.cpu().contiguous().view(torch.uint8)
        parts = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(parts, src, group=self.group)
        table.copy_(torch.cat(parts, 0).view(table.dtype).view(table.shape))
        if self.comm is not None:
            self.comm.host_ms += (time.perf_counter() - t0) * 1e3
        return GatheredTable(table, None)

    def all_to_all_v(self, out: torch.Tensor, inp: torch.Tensor, recv_splits, send_splits, async_op=False):
        """Variable all-to-all along dim 0.  Returns `out` (or a work handle when
        async_op under RCCL).  gloo has no all_to_all_single: host-staged isend/irecv."""
        if self.world == 1 and not self.active:
            out.copy_(inp)
            return None if async_op else out
        if self._backend() == "nccl":
            work = dist.all_to_all_single(out, inp.contiguous(), list(recv_splits), list(send_splits),
                                          group=self.group, async_op=async_op)
            return work if async_op else out
        t0 = time.perf_counter()
        raw = out.dtype == torch.bfloat16 and inp.dim() >= 2        # gloo point-to-point: move bf16 rows as bytes
        src = inp.contiguous().cpu()
        dst = torch.empty(out.shape, dtype=out.dtype)
        if raw:
            src, dst = src.view(torch.uint8), dst.view(torch.uint8)
        soff, roff = [0], [0]
        for r in range(self.world):
            soff.append(soff[-1] + int(send_splits[r]))
            roff.append(roff[-1] + int(recv_splits[r]))
        dst[roff[self.rank]:roff[self.rank + 1]] = src[soff[self.rank]:soff[self.rank + 1]]
        # pairwise and blocking, peers in increasing order, the lower rank of a pair sends first: the
        # smallest unfinished pair can always proceed, so no message size can deadlock it (posting
        # every isend before any recv did, at 4 ranks with multi-MB id lists)
        for peer in range(self.world):
            if peer == self.rank:
                continue
            for phase in (0, 1):
                if (phase == 0) == (self.rank < peer):
                    if soff[peer + 1] > soff[peer]:
                        dist.send(src[soff[peer]:soff[peer + 1]].contiguous(), peer, group=self.group)
                elif roff[peer + 1] > roff[peer]:
                    buf = torch.empty((roff[peer + 1] - roff[peer],) + tuple(dst.shape[1:]), dtype=dst.dtype)
                    dist.recv(buf, peer, group=self.group)
                    dst[roff[peer]:roff[peer + 1]] = buf
        out.copy_(dst.view(out.dtype) if raw else dst)
        if self.comm is not None:
            self.comm.host_ms += (time.perf_counter() - t0) * 1e3
        return None if async_op else out

    def plan_exchange(self, g_local: CSRGraph, max_halo_fraction: float = 0.6):
        """HaloPlan for `g_local` (a row block with GLOBAL colidx), or None when the
        halo would be more than `max_halo_fraction` of all remote rows (-> all-gather).
        Collective: every rank must call it for the same graph in the same order."""
        plan = HaloPlan(self, g_local)
        frac = torch.tensor([plan.halo_fraction], dtype=torch.float64,
                            device=g_local.device if self._backend() == "nccl" else "cpu")
        dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=self.group)
        return plan if float(frac.item()) <= max_halo_fraction else None

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        if not self.active:
            return flat
        ev = None
        if self.comm is not None:
            self.comm.allreduce_bytes += flat.numel() * flat.element_size()
            ev = self.comm.bracket(flat.device)
        if ev is not None:
            ev[0].record()
        if self._backend() == "nccl" or not flat.is_cuda:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if ev is not None:
                ev[1].record()
        else:
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host)
            if ev is not None:
                ev[1].record()
        return flat


class MaskedBackwardPlan:
    """Opt-in masked backward (HANTrainer(masked_backward=True)) of ONE meta-path: with a single node-attention
    layer the g row of every destination outside the loss mask is identically zero, so those destinations add
    exactly 0 to dH_j and df2_j of their sources.  The transposed graph keeps their entries as -1 IN PLACE (the
    kernel skips them and sums the rest in the order of the full pass: bit-identical results); under a node
    partition only the LIVE rows of the backward table [g | stats] travel -- every rank contributes its live rows
    (padded to the largest count) to one all-gather, 384 B x |mask| instead of 384 B x N per meta-path.

    graph_t   masked transposed shard (rows = local sources; entries = rows of `table`, or -1)
    live_idx  local rows (int64) that are live, or None on one GPU (the local table is used as it is)
    gid       (world * per_rank,) int32 global id of each row of the gathered table (dropout RNG keys), or None
    """

    def __init__(self, part: "NodePartition | None", graph_t: CSRGraph, live_local: torch.Tensor,
                 live_global: torch.Tensor | None = None):
        dev = graph_t.device
        self.part = part if (part is not None and part.active) else None
        if self.part is None:
            self.graph_t = graph_t.with_masked_columns(live_local.bool())
            self.live_idx, self.gid, self.per_rank = None, None, int(live_local.numel())
            self.rows_on_wire = 0
            return
        pt = self.part
        lg = live_global.bool()                                    # (n_table,) in global numbering
        counts = lg.view(pt.world, pt.shard).sum(1)
        self.per_rank = max(int(counts.max()), 1)
        # position of every live node among its owner's live rows -> its row in the gathered compact table
        pos = torch.cumsum(lg.view(pt.world, pt.shard).long(), 1) - 1
        remap = (pos + torch.arange(pt.world, device=dev)[:, None] * self.per_rank).view(-1)
        self.graph_t = graph_t.with_masked_columns(lg, remap, n_cols=pt.world * self.per_rank)
        self.live_idx = torch.nonzero(live_local.bool()).flatten()
        gid = torch.full((pt.world * self.per_rank,), 0, dtype=torch.int32, device=dev)
        ids = torch.nonzero(lg).flatten()
        gid[remap[ids]] = ids.to(torch.int32)
        self.gid = gid
        self.rows_on_wire = (pt.world - 1) * self.per_rank        # rows this rank receives per meta-path and step

    def table_async(self, gs_local: torch.Tensor, tag):
        """The table the masked transposed graph indexes: the local table itself on one GPU, else the all-gather of
        every rank's live rows (started asynchronously; `.wait()` yields it)."""
        if self.part is None:
            return GatheredTable(gs_local, None)
        packed = gs_local.index_select(0, self.live_idx)
        return self.part.all_gather_rows_async(packed, tag, rows_per_rank=self.per_rank)


class HaloPlan:
    """Exchange plan of one sharded graph: which remote table rows this rank reads.

    graph    the rank's row block with colidx REMAPPED into the [local | halo] table
    gid      (n_local + n_halo,) int32 global id of each table row (RNG keys)
    send_idx local rows this rank packs for the others, grouped by destination rank
    """

    def __init__(self, part: "NodePartition", g_local: CSRGraph):
        self.part = part
        dev = g_local.device
        shard, rank, world = part.shard, part.rank, part.world
        seen = torch.zeros(part.n_table, dtype=torch.bool, device=dev)       # bitmap instead of a sort-based unique
        seen[g_local.colidx.long()] = True
        needed = torch.nonzero(seen).flatten()                             # sorted global ids
        del seen
        owner = needed // shard
        remote = needed[owner != rank]
        recv_counts = torch.bincount(owner[owner != rank], minlength=world)
        send_counts = part.all_to_all_v(recv_counts.new_empty(world), recv_counts, [1] * world, [1] * world)
        self.recv_splits = [int(c) for c in recv_counts.tolist()]
        self.send_splits = [int(c) for c in send_counts.tolist()]
        req = remote.new_empty(sum(self.send_splits))
        part.all_to_all_v(req, remote, self.send_splits, self.recv_splits)   # ids the others want from me
        self.send_idx = (req - part.row_start).contiguous()
        self.n_local, self.n_halo = part.n_local, int(remote.numel())
        # remap colidx: local ids -> [0, n_local), remote ids -> n_local + position in `remote`
        c = g_local.colidx.long()
        is_local = (c >= part.row_start) & (c < part.row_end)
        pos = torch.searchsorted(remote, c.clamp(max=int(remote[-1]) if remote.numel() else 0)) \
            if remote.numel() else torch.zeros_like(c)
        new_col = torch.where(is_local, c - part.row_start, pos + self.n_local).to(torch.int32)
        self.graph = CSRGraph(g_local.rowptr, new_col.contiguous(), self.n_local + self.n_halo, validate=False,
                              values=g_local.values)
        local_ids = torch.arange(part.row_start, part.row_end, device=dev, dtype=torch.int64)
        self.gid = torch.cat([local_ids, remote]).to(torch.int32).contiguous()
        self.remote_rows_total = part.n_global - part.n_local

    @property
    def halo_fraction(self) -> float:
        return self.n_halo / max(self.remote_rows_total, 1)

    def exchange_async(self, local: torch.Tensor, tag=None) -> "GatheredTable":
        """local (n_local, ...) -> [local | halo] table (n_local + n_halo, ...)."""
        tail = tuple(local.shape[1:])
        part = self.part
        table = part.buffer(tag, (self.n_local + self.n_halo,) + tail, local.dtype, local.device)
        table[:self.n_local] = local
        packed = part.buffer(None if tag is None else (tag, "pack"), (self.send_idx.numel(),) + tail,
                             local.dtype, local.device)
        torch.index_select(local, 0, self.send_idx, out=packed)
        if part.comm is not None:
            part.comm.received(self.n_halo * (table[0].numel() if table.shape[0] else 0) * table.element_size())
        work = part.all_to_all_v(table[self.n_local:], packed, self.recv_splits, self.send_splits,
                                 async_op=True)
        return GatheredTable(table, work, keep=packed, comm=part.comm)


class GatheredTable:
    """Handle of an (possibly in-flight) all-gathered table."""

    def __init__(self, table, work, keep=None, comm: "CommStats | None" = None):
        self.table, self.work, self.keep, self.comm = table, work, keep, comm

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            ev = self.comm.bracket(self.table.device) if self.comm is not None else None
            if ev is not None:
                ev[0].record()
            self.work.wait()          # the current stream waits for the collective
            if ev is not None:
                ev[1].record()
            self.work, self.keep = None, None
        return self.table


# Replicated projection ("recompute instead of communicate").  xGMI is point-to-point: each pair of
# GPUs shares ONE link (about 60 GB/s per direction in practice), so receiving the (G-1)/G remote
# rows of a 256-B-per-row table costs rows * 256 B / (G * 60 GB/s) even when all G-1 links run in
# parallel, whereas K1 recomputes a row from its features at ~1.4 TB/s of feature bytes (0.74 ms per
# 1 GB of features with dropout, 0.40 ms without).  Per forward table at the
# SYN-1M shape (1M rows, F = 256): exchange 2.1 / 1.1 / 0.5 ms at G = 2 / 4 / 8 against 0.37 / 0.55 /
# 0.65 ms (training) and 0.20 / 0.30 / 0.35 ms (eval) of extra projection.  Hence: up to 4 ranks both
# forwards project the whole table; beyond that, where exchange and compute are about balanced, only
# the cheaper eval projection is replicated (it takes four of the twelve tables of an epoch off the
# links).  The backward table [g | stats] depends on the loss of its rows and is always exchanged.
def replication_policy(world: int, mode: str = "auto") -> frozenset:
    """Which forward passes project the whole table on every rank: subset of {"train", "eval"}."""
    mode = os.environ.get("HAN_REPLICATE", mode)
    if mode == "none" or world <= 1:
        return frozenset()
    if mode == "all":
        return frozenset(("train", "eval"))
    if mode == "eval":
        return frozenset(("eval",))
    if mode != "auto":
        raise ValueError(f"replicate = {mode!r}: expected auto / all / eval / none")
    return frozenset(("train", "eval")) if world <= 4 else frozenset(("eval",))


def _row_block(g: CSRGraph, r0: int, r1: int, n_cols: int) -> CSRGraph:
    rp = g.rowptr[r0:r1 + 1]
    s, e = int(rp[0]), int(rp[-1])
    return CSRGraph((rp - rp[0]).contiguous(), g.colidx[s:e].contiguous(), n_cols, validate=False,
                    values=g.values[s:e].contiguous() if g.values is not None else None, row_base=r0)
