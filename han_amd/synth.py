"""Synthetic HAN workloads of BASELINE.json's shapes.

There are no datasets in the container or on the GPU box (ACM3025.mat etc. are
external downloads), so every configuration is generated: random graphs with the
degree statistics of the named dataset, Gaussian features, uniform labels.
"""
from __future__ import annotations

import numpy as np
import torch

from .graph import CSRGraph


ROW_BLOCK = 1 << 16      # generation granule: every block of rows has its own generator seed


def _block_generator(seed: int, stream: int, block: int, device) -> torch.Generator:
    """Generator of row block `block` of random stream `stream`: what a block holds does not depend on
    which rows the caller asks for, so a rank of a node partition generates exactly its own rows
    (the global graph / feature matrix never exists on any rank) and N ranks together hold, bit for
    bit, what one process generates."""
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * 1_000_003 + int(stream)) * 2_097_169 + int(block))
    return g


def _row_range(n, rows):
    r0, r1 = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    if not (0 <= r0 <= r1 <= n):
        raise ValueError(f"rows {rows} outside [0, {n}]")
    return r0, r1


def _blocks(r0, r1):
    """(block id, first row, last row + 1) -- the rows as offsets inside the block -- for the 65 536-row
    blocks overlapping [r0, r1).  (Block-seeded generation dates from round 2: bench lines of round 1 were
    produced by a different generator and are not comparable row for row.)"""
    for b in range(r0 // ROW_BLOCK, (r1 + ROW_BLOCK - 1) // ROW_BLOCK if r1 > r0 else r0 // ROW_BLOCK):
        lo, hi = b * ROW_BLOCK, (b + 1) * ROW_BLOCK
        yield b, max(lo, r0) - lo, min(hi, r1) - lo


def random_regular_graph(n: int, deg: int, seed: int, device="cpu", symmetric=False, rows=None) -> CSRGraph:
    """Each row: the self-loop + (deg-1) DISTINCT uniformly random off-diagonal neighbours, columns sorted
    (SURVEY.md section 8d's recipe; rounds 1-3 kept the ~deg^2/2n repeated draws as multigraph terms).  Generated on
    `device` with torch so that 5e7-edge graphs take seconds: candidates are drawn from [0, n-1) and shifted past
    the row's own id; the few rows that drew a neighbour twice (0.12 % at deg 50, n = 10^6) are re-drawn whole, from a
    generator keyed by the attempt, until none is left -- all per 65 536-row block, so what a block holds does not
    depend on which rows the caller asks for.
    rows = (r0, r1): only those destination rows (colidx stays global, n_cols = n)."""
    if symmetric:
        raise NotImplementedError
    if deg - 1 > n - 1:
        raise ValueError(f"{deg - 1} distinct off-diagonal neighbours do not exist among {n} nodes")
    r0, r1 = _row_range(n, rows)
    parts = []
    for b, lo, hi in _blocks(r0, r1):
        nb_rows = min(ROW_BLOCK, n - b * ROW_BLOCK)
        ids_all = torch.arange(b * ROW_BLOCK, b * ROW_BLOCK + nb_rows, device=device, dtype=torch.int32)[:, None]

        def draw(gen, ids):
            if (deg - 1) ** 2 > n - 1:     # small tables (a row repeats a draw with probability > 0.4): a random permutation's prefix
                c = torch.rand((ids.shape[0], n - 1), generator=gen, device=device).argsort(1)[:, :deg - 1].to(torch.int32)
            else:
                c = torch.randint(0, n - 1, (ids.shape[0], deg - 1), generator=gen, device=device, dtype=torch.int32)
            return torch.sort(c + (c >= ids).to(torch.int32), dim=1).values      # skip the row's own id

        nb = draw(_block_generator(seed, 1, b, device), ids_all)
        for attempt in range(64):
            dup = (nb[:, 1:] == nb[:, :-1]).any(1) if deg > 2 else torch.zeros(nb_rows, dtype=torch.bool, device=device)
            idx = torch.nonzero(dup).flatten()
            if idx.numel() == 0:
                break
            nb[idx] = draw(_block_generator(seed, 101 + attempt, b, device), ids_all[idx])
        else:
            raise RuntimeError("could not draw distinct neighbours (deg too close to n)")
        ids = ids_all[lo:hi]
        parts.append(torch.sort(torch.cat([ids, nb[lo:hi]], dim=1), dim=1).values.reshape(-1))
        del nb
    cols = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32, device=device)
    rowptr = torch.arange(0, (r1 - r0) * deg + 1, deg, device=device, dtype=torch.int64)
    return CSRGraph(rowptr, cols.contiguous(), n, validate=False)


def banded_graph(n: int, deg: int, window: int, seed: int, device="cpu", rows=None) -> CSRGraph:
    """Graph with locality (what a partitioner leaves of a real meta-path graph): row i has
    its self-loop + (deg-1) neighbours uniform in [i-window, i+window] (wrapping).  Under a
    contiguous node partition only ~2*window remote rows per rank are referenced, so the
    halo exchange (dist.HaloPlan) replaces the all-gather."""
    r0, r1 = _row_range(n, rows)
    parts = []
    for b, lo, hi in _blocks(r0, r1):
        nb_rows = min(ROW_BLOCK, n - b * ROW_BLOCK)
        off = torch.randint(-window, window + 1, (nb_rows, deg - 1), generator=_block_generator(seed, 2, b, device),
                            device=device, dtype=torch.int32)[lo:hi]
        ids = torch.arange(b * ROW_BLOCK + lo, b * ROW_BLOCK + hi, device=device, dtype=torch.int32)[:, None]
        parts.append(torch.sort(torch.cat([ids, torch.remainder(ids + off, n)], dim=1), dim=1).values.reshape(-1))
        del off
    cols = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32, device=device)
    rowptr = torch.arange(0, (r1 - r0) * deg + 1, deg, device=device, dtype=torch.int64)
    return CSRGraph(rowptr, cols.contiguous(), n, validate=False)


def powerlaw_graph(n: int, nnz: int, alpha: float, seed: int, device="cpu", rows=None) -> CSRGraph:
    """Skewed variant: row degrees ~ Zipf-like with exponent alpha, scaled to
    about `nnz` edges, every row keeps its self-loop; neighbours uniform.  (The degree vector is
    drawn for all n rows on the host -- n numbers --, the edges only for the requested rows.)"""
    r0, r1 = _row_range(n, rows)
    rng = np.random.default_rng(seed)
    w = rng.pareto(alpha - 1.0, size=n) + 1.0
    deg = np.maximum(1, np.floor(w / w.sum() * nnz)).astype(np.int64)
    deg = np.minimum(deg, n)
    rowptr_g = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr_g[1:])
    parts = []
    for b, lo, hi in _blocks(r0, r1):
        b0 = b * ROW_BLOCK
        b1 = min(b0 + ROW_BLOCK, n)
        e_blk = int(rowptr_g[b1] - rowptr_g[b0])
        c = torch.randint(0, n, (e_blk,), generator=_block_generator(seed, 3, b, device), device=device,
                          dtype=torch.int32)
        parts.append(c[int(rowptr_g[b0 + lo] - rowptr_g[b0]):int(rowptr_g[b0 + hi] - rowptr_g[b0])])
    cols = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32, device=device)
    rp = torch.as_tensor(rowptr_g[r0:r1 + 1] - rowptr_g[r0], device=device)
    if r1 > r0:
        cols[rp[:-1]] = torch.arange(r0, r1, device=device, dtype=torch.int32)   # self-loop first
    return CSRGraph(rp, cols.contiguous(), n, validate=False)


def _symmetric_pairs(rng, n: int, density: float | None, nnz: int | None):
    """Strict-upper-triangle pairs (i < j) of a symmetric graph on n nodes: Bernoulli(density) per pair, or --
    nnz given -- EXACTLY (nnz - n) / 2 pairs drawn without replacement, so that the mirrored graph plus the n
    self-loops has exactly nnz entries (the data sets' edge counts, SURVEY.md section 8, include the self-loops)."""
    iu, ju = np.triu_indices(n, 1)
    if nnz is not None:
        if (nnz - n) % 2 or not (0 <= (nnz - n) // 2 <= iu.size):
            raise ValueError(f"a symmetric graph with self-loops on {n} nodes cannot have {nnz} entries")
        sel = rng.choice(iu.size, size=(nnz - n) // 2, replace=False)
    else:
        sel = np.nonzero(rng.random(iu.size) < density)[0]
    return iu[sel], ju[sel]


def bernoulli_graph(n: int, density: float | None, seed: int, device="cpu", nnz: int | None = None) -> CSRGraph:
    """ACM/DBLP-like: symmetric random edges + I (dense generation, n <= ~10k).  The strict upper triangle is
    drawn at `density` (every off-diagonal pair is an edge with that probability -- round 3 drew BOTH triangles at
    density/2 and OR-ed them, which gives d - d^2/4, 16 % short on a 78 % dense graph) or, with `nnz`, at exactly
    (nnz - n)/2 pairs; then mirrored, then the self-loops (utils/process.py:18-20)."""
    rng = np.random.default_rng(seed)
    i, j = _symmetric_pairs(rng, n, density, nnz)
    a = np.zeros((n, n), dtype=bool)
    a[i, j] = True
    a[j, i] = True
    np.fill_diagonal(a, True)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(a.sum(1), out=rowptr[1:])
    colidx = np.nonzero(a)[1].astype(np.int32)
    return CSRGraph.from_arrays(rowptr, colidx, n, device=device)


def planted_partition(n: int, c: int, p_metapaths: int, f: int, deg_in: int, deg_out: int, noise: float,
                      seed: int, device="cpu"):
    """A LEARNABLE synthetic task (the benchmark workloads have random labels): c communities;
    each meta-path graph links a node to `deg_in` random members of its own community and
    `deg_out` random other nodes (+ the self-loop); features = community one-hot (first c
    columns) + Gaussian noise.  Returns dict like make_workload (train 20 % / val 20 % / rest test)."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, c, n)
    members = [np.nonzero(labels == k)[0] for k in range(c)]
    graphs = []
    for q in range(p_metapaths):
        rows = []
        for i in range(n):
            same = rng.choice(members[labels[i]], size=deg_in)
            other = rng.integers(0, n, size=deg_out + 3 * q)     # later meta-paths are noisier
            rows.append(np.unique(np.concatenate([[i], same, other])))
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(r) for r in rows], out=rowptr[1:])
        graphs.append(CSRGraph.from_arrays(rowptr, np.concatenate(rows).astype(np.int32), n, device=device))
    x = np.zeros((n, f), dtype=np.float32)
    x[np.arange(n), labels] = 1.0
    x += noise * rng.standard_normal((n, f)).astype(np.float32)
    u = rng.random(n)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=device)
    return dict(x=t(x, torch.float32), graphs=graphs, labels=t(labels, torch.int32),
                train_mask=t(u < 0.2, torch.uint8), val_mask=t((u >= 0.2) & (u < 0.4), torch.uint8),
                test_mask=t(u >= 0.4, torch.uint8), n=n, f=f, c=c, p=p_metapaths, name="planted")


CONFIGS = {
    # name: (N, P, F, C, per-meta-path graph spec)
    # the data sets' entry counts incl. self-loops (PAP / PSP; APA / APCPA / APTPA), exactly
    "acm-like": dict(n=3025, f=1870, c=3, graphs=[("symmetric_nnz", 29281), ("symmetric_nnz", 2210761)]),
    "dblp-like": dict(n=4057, f=334, c=4, graphs=[("symmetric_nnz", 11113), ("symmetric_nnz", 5000495),
                                                  ("symmetric_nnz", 12924399)]),
    "syn-1m": dict(n=1_000_000, f=256, c=4, graphs=[("regular", 50)] * 4),
    "syn-1m-skew": dict(n=1_000_000, f=256, c=4, graphs=[("powerlaw", 50_000_000, 2.1)] * 4),
    # locality: neighbours within +-20000 of the row id -> halo exchange under a node partition
    "syn-1m-local": dict(n=1_000_000, f=256, c=4, graphs=[("banded", 50, 20_000)] * 4),
    # the same locality hidden behind arbitrary node ids (what a real data set looks like): the banded graphs
    # under ONE random relabelling of the nodes -- the input of the locality pass (han_amd.reorder)
    "syn-1m-local-shuffled": dict(n=1_000_000, f=256, c=4, graphs=[("banded_shuffled", 50, 20_000)] * 4),
    # tighter communities (neighbours within +-2000 of the row id), ids shuffled: after the locality pass the
    # source window of the rows one XCD has in flight fits its 4 MB L2
    "syn-1m-comm-shuffled": dict(n=1_000_000, f=256, c=4, graphs=[("banded_shuffled", 50, 2_000)] * 4),
    "syn-100k": dict(n=100_000, f=256, c=4, graphs=[("regular", 50)] * 4),
    # BASELINE.json configs[4] (meant for 8 GPUs; fits one 288 GB MI355X with bf16 tables)
    "syn-10m": dict(n=10_000_000, f=256, c=4, graphs=[("regular", 50)] * 8),
    "tiny": dict(n=512, f=48, c=3, graphs=[("bernoulli", 0.02), ("bernoulli", 0.2)]),
}


def make_graph(spec, n, seed, device, rows=None):
    kind = spec[0]
    if kind == "regular":
        return random_regular_graph(n, spec[1], seed, device, rows=rows)
    if kind == "powerlaw":
        return powerlaw_graph(n, spec[1], spec[2], seed, device, rows=rows)
    if kind == "banded":
        return banded_graph(n, spec[1], min(spec[2], max(n // 4, 1)), seed, device, rows=rows)
    if kind == "banded_shuffled":
        if rows is not None:
            raise NotImplementedError("the shuffled-ids workload is generated whole (single process)")
        from .reorder import permute_graph
        g = banded_graph(n, spec[1], min(spec[2], max(n // 4, 1)), seed, device)
        gen = torch.Generator(device=device)
        gen.manual_seed(424242 + n)                  # one relabelling for all meta-paths of a workload
        return permute_graph(g, torch.randperm(n, generator=gen, device=device))
    if kind in ("bernoulli", "symmetric_nnz"):
        if kind == "bernoulli":
            g = bernoulli_graph(n, spec[1], seed, device)
        else:      # a sample of the data set at another n (n_override) keeps the density
            cfg_n = next(c["n"] for c in CONFIGS.values() if spec in c["graphs"])
            nnz = spec[1] if n == cfg_n else n + 2 * int(round((spec[1] - cfg_n) / (cfg_n * (cfg_n - 1)) * n * (n - 1) / 2))
            g = bernoulli_graph(n, None, seed, device, nnz=nnz)
        if rows is None:
            return g
        from .dist import _row_block
        return _row_block(g, int(rows[0]), int(rows[1]), n)
    raise ValueError(kind)


def features(name: str, device="cpu", n_override: int | None = None, rows=None) -> torch.Tensor:
    """The feature rows of make_workload alone (same block generators): what a rank of a node partition
    generates when it wants the features of ALL rows for the replicated projection."""
    cfg = CONFIGS[name]
    n = int(n_override) if n_override else cfg["n"]
    r0, r1 = _row_range(n, rows)
    xs = []
    for b, lo, hi in _blocks(r0, r1):
        nb_rows = min(ROW_BLOCK, n - b * ROW_BLOCK)
        g = _block_generator(7, 0, b, device)
        xs.append(torch.randn((nb_rows, cfg["f"]), generator=g, device=device, dtype=torch.float32)[lo:hi])
    return torch.cat(xs) if xs else torch.empty((0, cfg["f"]), device=device)


def make_workload(name: str, device="cpu", seed: int = 1234, n_override: int | None = None, rows=None):
    """Returns dict(x (N,F) fp32, graphs [P CSRGraph], labels int32 (N,),
    train_mask / val_mask uint8 (N,), n, f, c, p).
    rows = (r0, r1): generate ONLY those rows of every tensor and graph (a rank's shard of a node
    partition; graphs keep global column ids); block-seeded, so the shards of N ranks concatenate
    to exactly what rows=None generates."""
    cfg = CONFIGS[name]
    n = int(n_override) if n_override else cfg["n"]
    r0, r1 = _row_range(n, rows)
    graphs = [make_graph(s, n, seed + p, device, rows=rows) for p, s in enumerate(cfg["graphs"])]
    xs, ls, us = [], [], []
    for b, lo, hi in _blocks(r0, r1):
        nb_rows = min(ROW_BLOCK, n - b * ROW_BLOCK)
        g = _block_generator(7, 0, b, device)
        xs.append(torch.randn((nb_rows, cfg["f"]), generator=g, device=device, dtype=torch.float32)[lo:hi])
        ls.append(torch.randint(0, cfg["c"], (nb_rows,), generator=g, device=device, dtype=torch.int32)[lo:hi])
        us.append(torch.rand((nb_rows,), generator=g, device=device)[lo:hi])
    x = torch.cat(xs) if xs else torch.empty((0, cfg["f"]), device=device)
    labels = torch.cat(ls) if ls else torch.empty(0, dtype=torch.int32, device=device)
    u = torch.cat(us) if us else torch.empty(0, device=device)
    train_mask = (u < 0.10).to(torch.uint8)
    val_mask = ((u >= 0.10) & (u < 0.20)).to(torch.uint8)
    return dict(x=x, graphs=graphs, labels=labels, train_mask=train_mask, val_mask=val_mask,
                n=n, f=cfg["f"], c=cfg["c"], p=len(graphs), name=name, rows=(r0, r1))
