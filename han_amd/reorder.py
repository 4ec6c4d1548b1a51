"""Locality pass: relabel the nodes so that rows processed close together gather from rows stored (and
re-used) close together.

The node attention (K2) gathers one 256-B projected row per edge.  On a graph whose node ids carry no
structure every gather misses the 4 MB L2 of the XCD (4 % hits measured on the random SYN-1M graph), and
under a node partition every rank references nearly all remote rows (all-gather instead of a halo).
Real meta-path graphs (co-author, co-subject, ...) have community structure but arbitrary ids; a
breadth-first relabelling over the union of the meta-path graphs puts a node next to its neighbours, so
consecutive destination rows -- the kernels walk rows in id order, blocks round-robin over the XCDs --
share their sources while those are still cached, and a contiguous row block references few remote rows.

The pass is a pure relabelling: `perm[new] = old`.  Features, labels, masks and graphs are permuted in,
per-node outputs are permuted back with `unpermute`; parameters and every per-graph quantity (loss,
accuracy, gradients) are unchanged up to fp32 summation order.  Dropout masks are keyed by node id, so a
training run on the relabelled problem draws a different -- equally valid -- mask assignment.

The reference has no counterpart (it feeds dense N x N masks, utils/process.py:14-25).
"""
from __future__ import annotations

import torch

from .graph import CSRGraph


def _ragged_gather_index(starts: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """Positions [starts[i], starts[i] + lens[i]) for every i, concatenated."""
    total = int(lens.sum())
    out_ptr = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=lens.device)
    torch.cumsum(lens, 0, out=out_ptr[1:])
    seg = torch.repeat_interleave(starts - out_ptr[:-1], lens)
    return seg + torch.arange(total, device=lens.device)


def permute_graph(g: CSRGraph, perm: torch.Tensor, inv: torch.Tensor | None = None) -> CSRGraph:
    """The square graph `g` under the relabelling perm[new] = old: row new = old row perm[new], its
    column ids mapped old -> new and sorted."""
    if g.n_rows != g.n_cols or perm.numel() != g.n_rows:
        raise ValueError("permute_graph needs a square graph and a permutation of its nodes")
    perm = perm.to(device=g.device, dtype=torch.int64)
    if inv is None:
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=g.device)
    deg = g.degrees()[perm]
    rowptr = torch.zeros(g.n_rows + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(deg, 0, out=rowptr[1:])
    src = _ragged_gather_index(g.rowptr[:-1][perm], deg)
    cols = inv[g.colidx[src].long()]
    vals = g.values[src] if g.values is not None else None
    # sort the columns inside each row (keeps the kernels' coalesced index loads monotone)
    rows = torch.repeat_interleave(torch.arange(g.n_rows, device=g.device), deg)
    order = torch.sort(rows * g.n_rows + cols).indices
    return CSRGraph(rowptr, cols[order].to(torch.int32).contiguous(), g.n_cols, validate=False,
                    values=vals[order].contiguous() if vals is not None else None)


def bfs_order(graphs, start: int | None = None) -> torch.Tensor:
    """Breadth-first (Cuthill-McKee style) order over the union of the meta-path graphs: perm[new] = old.
    Levels are expanded on the graphs' device with tensor ops; inside a level nodes keep the order of
    their first discoverer (neighbours of one node stay adjacent), components are taken in order of their
    smallest unvisited id.  Nodes without any neighbour but themselves (real meta-path graphs have many) are
    not walked one by one: they are appended at the end in id order, as one batch; and the next component's
    seed is found with a monotone cursor over the ids (one small slice per lookup), not a pass over all n."""
    g0 = graphs[0]
    n, dev = g0.n_rows, g0.device
    ids = torch.arange(n, device=dev)
    has_nb = torch.zeros(n, dtype=torch.bool, device=dev)
    for g in graphs:
        rows = torch.repeat_interleave(ids, g.degrees())
        has_nb[rows[g.colidx.long() != rows]] = True
    isolated = torch.nonzero(~has_nb).flatten()
    visited = ~has_nb                       # isolated nodes are placed at the end
    order = []
    done = int(isolated.numel())
    cursor, chunk = 0, 1 << 16
    next_seed = 0 if start is None else int(start)
    while done < n:
        if visited[next_seed]:
            while True:                     # smallest unvisited id >= cursor (ids below it are all visited)
                rest = torch.nonzero(~visited[cursor:cursor + chunk]).flatten()
                if rest.numel():
                    next_seed = cursor + int(rest[0])
                    cursor = next_seed
                    break
                cursor += chunk
        frontier = torch.tensor([next_seed], dtype=torch.int64, device=dev)
        visited[frontier] = True
        while frontier.numel():
            order.append(frontier)
            done += frontier.numel()
            cand = []
            for g in graphs:
                lens = g.degrees()[frontier]
                idx = _ragged_gather_index(g.rowptr[:-1][frontier], lens)
                cand.append(g.colidx[idx].long())
            c = torch.cat(cand)
            c = c[~visited[c]]
            if c.numel() == 0:
                break
            # first occurrence order: stable unique
            uniq, inverse = torch.unique(c, return_inverse=True)
            first = torch.full((uniq.numel(),), c.numel(), dtype=torch.int64, device=dev)
            first.scatter_reduce_(0, inverse, torch.arange(c.numel(), device=dev), reduce="amin")
            frontier = uniq[torch.sort(first).indices]
            visited[frontier] = True
    order.append(isolated)
    return torch.cat(order)


class Relabelled:
    """A workload (dict as han_amd.synth.make_workload returns) under a node relabelling."""

    def __init__(self, wl: dict, perm: torch.Tensor):
        dev = wl["x"].device
        self.perm = perm.to(dev)
        self.inv = torch.empty_like(self.perm)
        self.inv[self.perm] = torch.arange(self.perm.numel(), device=dev)
        self.wl = dict(wl)
        for k in ("x", "labels", "train_mask", "val_mask", "test_mask"):
            if k in wl and wl[k] is not None:
                self.wl[k] = wl[k][self.perm].contiguous()
        self.wl["graphs"] = [permute_graph(g, self.perm, self.inv) for g in wl["graphs"]]

    def unpermute(self, t: torch.Tensor, dim: int = 0) -> torch.Tensor:
        """Per-node output of the relabelled problem -> original node order."""
        return t.index_select(dim, self.inv)


def relabel(wl: dict, method: str = "bfs") -> Relabelled:
    if method != "bfs":
        raise ValueError(f"unknown reordering {method!r}")
    return Relabelled(wl, bfs_order(wl["graphs"]))


def halo_fraction(graphs, world: int) -> list[float]:
    """Per meta-path: the largest fraction of remote rows any rank of a contiguous `world`-way node
    partition references (what dist.HaloPlan would exchange; > 0.6 means all-gather)."""
    out = []
    for g in graphs:
        n = g.n_rows
        shard = (n + world - 1) // world
        worst = 0.0
        for r in range(world):
            r0, r1 = min(r * shard, n), min((r + 1) * shard, n)
            cols = g.colidx[int(g.rowptr[r0]):int(g.rowptr[r1])].long()
            seen = torch.zeros(n, dtype=torch.bool, device=g.device)
            seen[cols] = True
            seen[r0:r1] = False
            worst = max(worst, float(seen.sum()) / max(n - (r1 - r0), 1))
        out.append(worst)
    return out
