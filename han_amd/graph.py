"""Meta-path graphs in the layout the HIP kernels read: CSR with int64 row
pointers and int32 neighbour ids, plus the transposed (CSC) form the backward
gathers over.

The reference never builds a sparse structure: it feeds a dense additive mask
``bias_mat`` (0 / -1e9, ``utils/process.py:14-25``) or, in the unused
``sp_attn_head`` (``utils/layers.py:85``), a ``tf.SparseTensor``.  Both are
accepted here and converted once.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

EDGE_THRESHOLD = -1e8   # an entry of bias_mat is an edge iff it is > -1e8


def _stream():
    return torch.cuda.current_stream().cuda_stream


class CSRGraph:
    """rows = destination nodes i, columns = source nodes j (alpha_ij weights H_j).

    rowptr (n_rows+1,) int64, colidx (nnz,) int32, both on `device`.
    `n_cols` is the number of rows of the tables that colidx indexes.
    `values` (nnz,) fp32 or None: the stored values of sp_attn_head's SparseTensor
    adj_mat, which scale the logits (utils/layers.py:95-96); None == binary.
    """

    def __init__(self, rowptr: torch.Tensor, colidx: torch.Tensor, n_cols: int | None = None,
                 validate: bool = True, values: torch.Tensor | None = None, row_base: int = 0):
        if rowptr.dtype != torch.int64 or colidx.dtype != torch.int32:
            raise ValueError("rowptr must be int64 and colidx int32")
        if rowptr.dim() != 1 or colidx.dim() != 1 or rowptr.numel() < 1:
            raise ValueError("rowptr/colidx must be 1-D")
        if rowptr.device != colidx.device:
            raise ValueError("rowptr and colidx must live on the same device")
        self.rowptr = rowptr.contiguous()
        self.colidx = colidx.contiguous()
        self.n_rows = rowptr.numel() - 1
        self.n_cols = int(n_cols) if n_cols is not None else self.n_rows
        self.nnz = colidx.numel()
        if values is not None:
            if values.shape != (self.nnz,) or values.device != colidx.device:
                raise ValueError("values must be (nnz,) on the graph's device")
            values = values.to(torch.float32).contiguous()
        self.values = values
        # id, in the numbering of the columns, of row 0: the row / transposed shards of a node partition hold
        # local rows [0, n_local) against global column ids (has_locality compares like with like)
        self.row_base = int(row_base)
        self._t = None
        self._split = {}
        self._locality = None
        self.masked = False         # True: colidx holds -1 for skipped entries (with_masked_columns)
        if validate:
            self.validate()

    # ---- checks -----------------------------------------------------------
    def validate(self):
        rp = self.rowptr
        if int(rp[0]) != 0 or int(rp[-1]) != self.nnz:
            raise ValueError("rowptr[0] must be 0 and rowptr[-1] == nnz")
        if self.n_rows > 0 and bool((rp[1:] < rp[:-1]).any()):
            raise ValueError("rowptr must be non-decreasing")
        if self.nnz:
            lo, hi = int(self.colidx.min()), int(self.colidx.max())
            if lo < 0 or hi >= self.n_cols:
                raise ValueError(f"colidx out of range [0,{self.n_cols}): [{lo},{hi}]")

    @property
    def device(self):
        return self.rowptr.device

    def degrees(self) -> torch.Tensor:
        return self.rowptr[1:] - self.rowptr[:-1]

    def has_locality(self) -> bool:
        """True when neighbour ids sit close to the row id (mean |col - row| below 5 % of the table): the K2
        launches then use the XCD-aware work order (HAN_FLAG_XCD_ORDER).  One pass over a sample of the
        edges, cached; graphs indexing a [local | halo] table (remapped columns) count as local."""
        if self._locality is None:
            if self.nnz == 0 or self.n_rows == 0:
                self._locality = False
            else:
                step = max(1, self.nnz // 2_000_000)
                pos = torch.arange(0, self.nnz, step, device=self.device)
                rows = torch.searchsorted(self.rowptr, pos, right=True) - 1
                dist = (self.colidx[pos].long() - (rows + self.row_base)).abs().double().mean()
                self._locality = bool(float(dist) < 0.05 * max(self.n_cols, 1))
        return self._locality

    def with_masked_columns(self, live: torch.Tensor, remap: torch.Tensor | None = None,
                            n_cols: int | None = None) -> "CSRGraph":
        """The same rows with every entry whose column is not `live` replaced by -1 IN PLACE (positions kept),
        the others optionally renumbered through `remap` (HAN_FLAG_MASKED_EDGES form of a transposed graph:
        han_node_attn_bwd_cols skips the negative entries and sums the rest in the order of the full graph)."""
        c = self.colidx.long()
        new = torch.where(live[c], c if remap is None else remap[c], torch.full_like(c, -1)).to(torch.int32)
        g = CSRGraph(self.rowptr, new.contiguous(), self.n_cols if n_cols is None else n_cols, validate=False,
                     values=self.values, row_base=self.row_base)
        g._locality = False         # its column ids are no longer row ids
        g.masked = True
        return g

    def has_empty_rows(self) -> bool:
        return self.n_rows > 0 and bool((self.degrees() == 0).any())

    def to(self, device) -> "CSRGraph":
        g = CSRGraph(self.rowptr.to(device), self.colidx.to(device), self.n_cols, validate=False,
                     values=self.values.to(device) if self.values is not None else None, row_base=self.row_base)
        return g

    # ---- row splitting for skewed degree distributions ---------------------------
    def row_split(self, split_deg: int = 1024, chunk: int = 512):
        """Rows longer than `split_deg` cut into chunks of `chunk` edges (the
        han_row_split_t description).  Returns None when no row is that long,
        else a dict of device tensors; cached per (split_deg, chunk)."""
        key = (int(split_deg), int(chunk))
        if key not in self._split:
            deg = self.degrees()
            long_rows = torch.nonzero(deg > split_deg).flatten()
            if long_rows.numel() == 0:
                self._split[key] = None
            else:
                ldeg = deg[long_rows]
                nch = (ldeg + chunk - 1) // chunk
                long_ptr = torch.zeros(long_rows.numel() + 1, dtype=torch.int64, device=self.device)
                torch.cumsum(nch, 0, out=long_ptr[1:])
                n_chunks = int(long_ptr[-1])
                chunk_long = torch.repeat_interleave(
                    torch.arange(long_rows.numel(), device=self.device, dtype=torch.int32), nch)
                within = torch.arange(n_chunks, device=self.device) - long_ptr[:-1][chunk_long.long()]
                base = self.rowptr[long_rows][chunk_long.long()]
                chunk_start = (base + within * chunk).contiguous()
                chunk_end = torch.minimum(chunk_start + chunk,
                                          self.rowptr[long_rows + 1][chunk_long.long()]).contiguous()
                self._split[key] = dict(split_deg=int(split_deg), n_long=int(long_rows.numel()),
                                        n_chunks=n_chunks, long_rows=long_rows.contiguous(),
                                        long_ptr=long_ptr, chunk_long=chunk_long.contiguous(),
                                        chunk_start=chunk_start, chunk_end=chunk_end)
        return self._split[key]

    def row_bins(self, short_deg: int = 16, split_deg: int = 1024):
        """Degree bins of the rows (han_row_split_t: short_rows / mid_rows): rows with fewer than `short_deg` entries
        (incl. empty rows), ordered by ceil(deg / 4) and then id -- four of them share a wave, one 16-lane group
        each, and should need the same number of 4-entry steps --, rows of short_deg .. split_deg entries in id order
        (a wave each), rows beyond split_deg (row_split's chunks).  Returns dict(n_short, short_rows, n_mid, mid_rows)
        with int32 device tensors; a list is None when its bin holds EVERY row (identity).  Cached."""
        key = ("bins", int(short_deg), int(split_deg))
        if key not in self._split:
            deg = self.degrees()
            is_short = deg < short_deg
            is_mid = (~is_short) & (deg <= split_deg)
            n_short, n_mid = int(is_short.sum()), int(is_mid.sum())
            short_rows = mid_rows = None
            if 0 < n_short < self.n_rows:
                ids = torch.nonzero(is_short).flatten()
                order = torch.sort((deg[ids] + 3) // 4, stable=True).indices
                short_rows = ids[order].to(torch.int32).contiguous()
            if 0 < n_mid < self.n_rows:
                mid_rows = torch.nonzero(is_mid).flatten().to(torch.int32).contiguous()
            self._split[key] = dict(n_short=n_short, short_rows=short_rows, n_mid=n_mid, mid_rows=mid_rows)
        return self._split[key]

    def bitmask(self):
        """Adjacency bit mask for the dense (matrix-pipe) K2 path: (n_rows, ceil(n_cols / 32)) int32 on the device, bit
        (j & 31) of word (j >> 5) of row i set iff the graph stores (i, j) -- or None when the graph holds repeated
        entries (a multigraph term has no bit-mask form) or lives on the host.  Built once by han_csr_to_bitmask, cached."""
        if "bitmask" not in self._split:
            bm = None
            if self.rowptr.is_cuda and self.n_rows > 0 and self.n_cols > 0 and not self.masked:
                lib = _lib.load()
                ldw = (self.n_cols + 31) // 32
                bits = torch.empty((self.n_rows, ldw), dtype=torch.int32, device=self.device)
                rep = torch.zeros(1, dtype=torch.int32, device=self.device)
                _lib.check(lib.han_csr_to_bitmask(self.rowptr.data_ptr(), self.colidx.data_ptr(), self.n_rows, self.n_cols,
                                                  bits.data_ptr(), ldw, rep.data_ptr(), _stream()), "han_csr_to_bitmask")
                if int(rep.item()) == 0:
                    bm = bits
            self._split["bitmask"] = bm
        return self._split["bitmask"]

    # ---- transposition (CSC) ------------------------------------------------
    def transpose(self) -> "CSRGraph":
        """The transposed graph: for every source j the destinations i, in
        ascending i (stable), as a CSRGraph with n_rows = n_cols of self."""
        if self._t is None:
            rows = torch.repeat_interleave(
                torch.arange(self.n_rows, device=self.device, dtype=torch.int32), self.degrees())
            cols = self.colidx.long()
            order = torch.sort(cols, stable=True).indices
            rowidx = rows[order].contiguous()
            counts = torch.bincount(cols, minlength=self.n_cols)
            colptr = torch.zeros(self.n_cols + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(counts, 0, out=colptr[1:])
            t = CSRGraph(colptr, rowidx, n_cols=self.n_rows, validate=False,
                         values=self.values[order] if self.values is not None else None)
            t._t = self
            self._t = t
        return self._t

    # ---- constructors ---------------------------------------------------------
    @staticmethod
    def from_bias(bias_mat: torch.Tensor) -> "CSRGraph":
        """Dense additive mask (N,N) or (1,N,N) -> CSR on the tensor's device.
        On a GPU tensor this runs the HIP count/fill kernels (the replacement
        for the O(N^2) Python loop of utils/process.py:21-24 + the dense feed)."""
        b = bias_mat
        if b.dim() == 3:
            if b.shape[0] != 1:
                raise ValueError("batch size must be 1 (as the reference, ex_acm3025.py:21)")
            b = b[0]
        if b.dim() != 2 or b.shape[0] != b.shape[1]:
            raise ValueError("bias_mat must be (N,N) or (1,N,N)")
        n = b.shape[0]
        if b.is_cuda:
            lib = _lib.load()
            b = b.to(torch.float32)
            if b.stride(1) != 1:
                b = b.contiguous()
            counts = torch.empty(n, dtype=torch.int64, device=b.device)
            _lib.check(lib.han_bias_row_counts(b.data_ptr(), n, b.stride(0), counts.data_ptr(),
                                               _stream()), "han_bias_row_counts")
            rowptr = torch.zeros(n + 1, dtype=torch.int64, device=b.device)
            torch.cumsum(counts, 0, out=rowptr[1:])
            nnz = int(rowptr[-1])
            colidx = torch.empty(nnz, dtype=torch.int32, device=b.device)
            _lib.check(lib.han_bias_fill_csr(b.data_ptr(), n, b.stride(0), rowptr.data_ptr(),
                                             colidx.data_ptr(), _stream()), "han_bias_fill_csr")
            return CSRGraph(rowptr, colidx, n, validate=False)
        keep = b > EDGE_THRESHOLD
        rowptr = torch.zeros(n + 1, dtype=torch.int64)
        torch.cumsum(keep.sum(1), 0, out=rowptr[1:])
        colidx = keep.nonzero()[:, 1].to(torch.int32)
        return CSRGraph(rowptr, colidx, n, validate=False)

    @staticmethod
    def from_adjacency(adj, add_self_loops: bool = True, device=None) -> "CSRGraph":
        """Dense/scipy adjacency (N,N): edge iff (adj + I)_ij > 0 -- the nhood=1
        case of utils/process.py:14-25 without materialising the mask."""
        import scipy.sparse as sp
        a = sp.csr_matrix(adj)
        a = (a != 0).astype(np.int8)
        if add_self_loops:
            a = ((a + sp.identity(a.shape[0], dtype=np.int8, format="csr")) != 0).astype(np.int8)
        a.sort_indices()
        return CSRGraph.from_arrays(a.indptr, a.indices, a.shape[1], device=device)

    @staticmethod
    def from_arrays(rowptr, colidx, n_cols=None, device=None) -> "CSRGraph":
        rp = torch.as_tensor(np.asarray(rowptr, dtype=np.int64))
        ci = torch.as_tensor(np.asarray(colidx, dtype=np.int32))
        if device is not None:
            rp, ci = rp.to(device), ci.to(device)
        return CSRGraph(rp, ci, n_cols)

    @staticmethod
    def from_torch_sparse(t: torch.Tensor) -> "CSRGraph":
        """A torch sparse COO/CSR tensor of shape (N,N) or (1,N,N) (the rank-3
        SparseTensor of utils/layers.py:85-115).  All-ones values give a binary
        graph; anything else is kept as `values` (they scale the logits, :95-96)."""
        def vals_of(v):
            return None if (v.numel() == 0 or bool(torch.all(v == 1))) else v.to(torch.float32)
        if t.layout == torch.sparse_csr:
            return CSRGraph(t.crow_indices().to(torch.int64), t.col_indices().to(torch.int32),
                            t.shape[-1], values=vals_of(t.values()))
        t = t.coalesce()
        idx = t.indices()
        if idx.shape[0] == 3:
            if t.shape[0] != 1:
                raise ValueError("batch size must be 1 (utils/layers.py:110-113)")
            idx = idx[1:]
        n = t.shape[-2]
        order = torch.argsort(idx[0] * t.shape[-1] + idx[1])
        rows, cols = idx[0][order], idx[1][order]
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=t.device)
        torch.cumsum(torch.bincount(rows, minlength=n), 0, out=rowptr[1:])
        return CSRGraph(rowptr, cols.to(torch.int32), t.shape[-1], values=vals_of(t.values()[order]))

    def to_bias(self, dtype=torch.float32) -> torch.Tensor:
        """Back to the reference's dense additive mask (1,N,N) -- tests only."""
        b = torch.full((self.n_rows, self.n_cols), -1e9, dtype=dtype, device=self.device)
        rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), self.degrees())
        b[rows, self.colidx.long()] = 0.0
        return b[None]


def as_graph(g, device=None) -> CSRGraph:
    """Accept a CSRGraph, a dense bias matrix, a torch sparse tensor or a
    (rowptr, colidx[, n_cols]) tuple -- the `bias_mat_list[p]` / `adj_mat`
    arguments of the reference surface."""
    if isinstance(g, CSRGraph):
        out = g
    elif isinstance(g, torch.Tensor) and g.layout != torch.strided:
        out = CSRGraph.from_torch_sparse(g)
    elif isinstance(g, torch.Tensor):
        if device is not None and g.device != torch.device(device):
            g = g.to(device)
        out = CSRGraph.from_bias(g)
    elif isinstance(g, np.ndarray):
        t = torch.as_tensor(g, dtype=torch.float32)
        out = CSRGraph.from_bias(t.to(device) if device is not None else t)
    elif isinstance(g, (tuple, list)) and len(g) in (2, 3):
        out = CSRGraph.from_arrays(*g, device=device) if not isinstance(g[0], torch.Tensor) \
            else CSRGraph(g[0], g[1], g[2] if len(g) == 3 else None)
    else:
        raise TypeError(f"cannot interpret {type(g)} as a meta-path graph")
    if device is not None and out.device != torch.device(device):
        out = out.to(device)
    return out
