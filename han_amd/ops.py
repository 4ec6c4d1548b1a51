"""Thin, validating Python wrappers over the C ABI (one per entry point).

Tensors are plumbing: device memory + the current HIP stream.  Every wrapper
checks device / dtype / contiguity / shape and raises ``ValueError`` before the
kernel sees the pointers (the kernels assume validated input).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .graph import CSRGraph

ACT_IDENTITY = 0
ACT_ELU = 1
FLAG_XCD_ORDER = 1         # HAN_FLAG_XCD_ORDER: XCD-aware work order for graphs with locality
FLAG_LEAN = 256            # HAN_FLAG_LEAN: the lean K2 kernels of small graphs (scores read from the table, shared dropout hash, one lane per head at 8 x 8)
FLAG_K2_DEEP = 512         # HAN_FLAG_K2_DEEP (measurements: bf16 eval forward with 8 steps in flight)
K2_DEEP = False            # set by tools/k2_regimes.py --deep
K2_SHARED_HASH = False     # tests / measurements: FLAG_K2_SHARED_HASH on every node_attn_fwd call
FLAG_MASKED_EDGES = 64     # HAN_FLAG_MASKED_EDGES: negative entries of the transposed graph are skipped in place
FLAG_K1_EXACT_PIPE = 2     # HAN_FLAG_K1_EXACT_PIPE / _MATRIX_PIPE: force one of the two K1 forward kernels (tests, measurements)
FLAG_K1_MATRIX_PIPE = 4
FLAG_K1_4WAVE = 16         # HAN_FLAG_K1_4WAVE (measurements: the two-waves-per-SIMD form of the bf16 x 6 kernel)
FLAG_K1_PAIRS = 32         # HAN_FLAG_K1_PAIRS (measurements: project_fwd_multi fuses 2 meta-paths per block, not 4)
FLAG_K3_EXACT_PIPE = 8     # HAN_FLAG_K3_EXACT_PIPE: fp32 MFMA K3 kernels also for large inputs
FLAG_K2_SHARED_HASH = 1024 # HAN_FLAG_K2_SHARED_HASH: measurements only (one attention-dropout hash per edge and four heads)
FLAG_K3_PAIRS = 16         # HAN_FLAG_K3_PAIRS: measurements only (two waves share a tile in the K3 backward)
FLAG_K3_G3_F32 = 32        # HAN_FLAG_K3_G3_F32: measurements only (dW product of the K3 backward on the fp32 pipe)


def flag_fts_slice(s: int) -> int:
    """HAN_FLAG_FTS_SLICE(s): column slice s of a head wider than 64 columns (project_fwd: the projected-row
    dropout of slice s draws from stream 2 + 4 s; everything else is keyed as for slice 0)."""
    if not 0 <= s < 256:
        raise ValueError("at most 256 slices of 64 columns per head")
    return (int(s) & 0xFF) << 8


LEAKY_SLOPE = 0.2          # tf.nn.leaky_relu default (utils/layers.py:27)
D = 64                     # K * F' of this build
STATS_ROW_BYTES = 128      # per destination row: the (f1, lse, s, 0) records of the K = 8 heads inside the fused gs row (bench.py byte model)

_workspaces: dict = {}
_retired_workspaces: list = []      # superseded buffers stay alive: a captured hipGraph may have baked their pointers in

# Optional timing hook (bench.py): a list to which node_attn_fwd / node_attn_bwd_cols
# append (tag, start_event, end_event, N, E) recorded on the launch stream.
K2_TIMING: list | None = None


def _dev_word(t):
    """Device pointer of a 1-element int64 tensor (seed_dev / step_dev of the C ABI), or None."""
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
        raise ValueError("seed_dev / step_dev must be a 1-element int64 GPU tensor")
    return t.data_ptr()


def _out(t, name, shape, dev):
    """A caller-provided destination (e.g. a slice of the flat gradient buffer) or a fresh tensor."""
    if t is None:
        return torch.empty(shape, dtype=torch.float32, device=dev)
    _chk(t, name, tuple(shape), device=dev)
    return t


def _stream():
    return torch.cuda.current_stream().cuda_stream


WS_SUFFIX = ""             # set by layers._on_path: kernels of different meta-paths that run concurrently (one side stream per
                           # meta-path inside a captured epoch) must not share a scratch buffer


def _ws(nbytes: int, device, tag: str = "") -> torch.Tensor:
    """Grow-only scratch buffer per (device, tag).  A buffer that has to grow is REPLACED, and the old
    one is kept alive for the life of the process (never returned to the caching allocator): an epoch
    captured into a hipGraph (HANTrainer(use_graph=True)) replays with the raw pointers it saw, and a
    later, larger request for the same tag must not free memory such a graph still reads and writes.
    Growth is geometric, so the retired buffers of a tag add up to less than its current one."""
    key = (str(device), tag + WS_SUFFIX)
    t = _workspaces.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _retired_workspaces.append(t)
        t = torch.empty(max(int(nbytes), 1 << 20, 2 * (t.numel() if t is not None else 0)), dtype=torch.uint8,
                        device=device)
        _workspaces[key] = t
    return t


def require_gpu(t: torch.Tensor, name: str) -> torch.Tensor:
    """Guard of the (few) torch-level paths around the kernels: no CPU path there either."""
    if not t.is_cuda:
        raise ValueError(f"{name}: must be a GPU tensor (han_amd has no CPU path); got {t.device}")
    return t


def _chk(t: torch.Tensor, name: str, shape=None, dtype=torch.float32, device=None, contiguous=True):
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must be a GPU tensor (han_amd has no CPU path); got {t.device}")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and int(a) != int(s)
                                        for a, s in zip(t.shape, shape)):
            raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}      # HAN_DTYPE_F32 / HAN_DTYPE_BF16


def _dtype_code(t: torch.Tensor, name: str) -> int:
    if t.dtype not in DTYPE_CODE:
        raise ValueError(f"{name}: dtype {t.dtype}, expected float32 or bfloat16")
    return DTYPE_CODE[t.dtype]


# rows longer than SPLIT_DEG are cut into chunks of SPLIT_CHUNK edges, a wave per chunk (round 4: 1024 / 512, from 8192 /
# 4096 -- a wave walking an 8192-entry row alone was the tail of the whole launch on power-law graphs; the finishing
# launch now merges a row's chunks with a block instead of one lane group: profiles/r04_k2_skew_split_sweep.jsonl)
SPLIT_DEG = 1024
SPLIT_CHUNK = 512


SHORT_DEG = 16         # HAN_SHORT_DEG: rows below this many entries run four to a wave in a degree-binned launch
BINNED = True          # tests / measurements: False = one row shape per launch, chosen from E / N (rounds 1-3)


def _row_split_arg(graph: CSRGraph, tag: str, bins: bool = True, split: bool = True):
    """ctypes han_row_split_t for `graph`: its degree bins (short / mid row lists) and the chunks of its rows beyond
    SPLIT_DEG (None when the graph has neither: every row in one bin and no long row still passes the bin counts,
    so that the library uses that bin's row shape).  Returns (byref-able struct or None, keepalive tuple)."""
    sp = graph.row_split(SPLIT_DEG, SPLIT_CHUNK) if split else None      # (the lean kernels walk whole rows)
    rb = graph.row_bins(SHORT_DEG, SPLIT_DEG) if (bins and BINNED and graph.n_rows > 0) else None
    if sp is None and rb is None:
        return None, None
    lib = _lib.load()
    ptr = lambda t: t.data_ptr() if t is not None else None
    if sp is not None:
        ws = _ws(lib.han_row_split_workspace(sp["n_chunks"]), graph.device, "split" + tag)
        st = _lib.HanRowSplit(sp["split_deg"], sp["n_long"], sp["n_chunks"], sp["long_rows"].data_ptr(),
                              sp["long_ptr"].data_ptr(), sp["chunk_long"].data_ptr(),
                              sp["chunk_start"].data_ptr(), sp["chunk_end"].data_ptr(), ws.data_ptr(),
                              ws.numel(), 0, 0, None, None)
    else:
        ws = None
        st = _lib.HanRowSplit(SPLIT_DEG, 0, 0, None, None, None, None, None, None, 0, 0, 0, None, None)
    if rb is not None:
        st.n_short, st.n_mid = rb["n_short"], rb["n_mid"]
        st.short_rows, st.mid_rows = ptr(rb["short_rows"]), ptr(rb["mid_rows"])
    return st, (sp, rb, ws)


LEAN = True                # tests / measurements: False keeps the classic gather kernels


def _use_lean(graph: CSRGraph, table) -> bool:
    """The lean K2 forward (HAN_FLAG_LEAN): tables that live in the L2s (<= 16384 rows: 4 MB) with long rows (mean
    degree >= 64), where the classic gather kernels are bound by vector-instruction issue, not by memory."""
    return (LEAN and table.dtype == torch.float32 and 0 < graph.n_cols <= 16384
            and graph.nnz >= 64 * graph.n_rows)


DENSE = True               # tests / measurements: False keeps small dense graphs on the lean CSR kernels
DENSE_MIN_DENSITY = 0.65   # stored entries / (rows x table rows) from which the matrix-pipe form wins over a whole training step (eval + training forward + backward: 0.47 ms at any density against 0.59 ms for the lean CSR kernels at 78 %, ~0.40 ms at 50 %; profiles/r04_k2_dense_vs_lean.jsonl)


def _use_dense(graph: CSRGraph, table, K: int, FP: int) -> bool:
    """The dense (bit mask + fp32 MFMA) K2 path: a lean-eligible graph (_use_lean) of the reference shape whose rows
    are dense enough that computing every (i, j) pair beats walking the stored entries, binary, without repeated
    entries."""
    if not (DENSE and K == 8 and FP == 8 and graph.values is None and not graph.masked and _use_lean(graph, table)):
        return False
    if graph.nnz < DENSE_MIN_DENSITY * graph.n_rows * graph.n_cols:
        return False
    return graph.bitmask() is not None


def _dense_arg(graph: CSRGraph, train: bool, tag: str):
    """ctypes han_dense_t of `graph` (+ keepalive)."""
    lib = _lib.load()
    bits = graph.bitmask()
    ws = _ws(lib.han_node_attn_dense_workspace(graph.n_rows, graph.n_cols, int(train)), graph.device, "dense" + tag)
    return _lib.HanDense(bits.data_ptr(), bits.shape[1], graph.n_cols, ws.data_ptr(), ws.numel()), (bits, ws)


def _check_heads(K: int, FP: int):
    if K * FP != D or FP not in (4, 8, 16, 32, 64):
        raise NotImplementedError(
            f"this build supports n_heads*hid_units == 64 with hid_units in "
            f"{{4,8,16,32,64}}; got n_heads={K}, hid_units={FP}")


def _check_drop(p: float, name: str):
    p = float(p)
    if not (0.0 <= p < 1.0):
        raise ValueError(f"{name} must be in [0, 1), got {p}")
    return p


# --------------------------------------------------------------------------- K1
def project_fwd(X, W, a1, a2, b1, b2, in_drop=0.0, fts_drop=0.0, seed=0, row_offset=0,
                table_dtype=torch.float32, seed_dev=None, flags=0, want_keep=False):
    """utils/layers.py:18-24,31-32 for the K heads of one meta-path.
    X (N,F) fp32 or bf16 [row stride >= F]; W (F,D); a1,a2 (K,F'); b1,b2 (K,).
    H is stored in `table_dtype` (float32 or bfloat16; f1/f2 come from the stored rows).
    Returns H (N,D), f1 (N,K), f2 (N,K).  With fts_drop > 0 the keep bit of the
    projected-row dropout rides in mantissa bit 0 of every H element.
    want_keep=True returns a 4th value: the keep table of the per-head input dropout for
    project_bwd (uint8, han_project_keep_bytes() long), or None when this shape has none
    (small inputs, in_drop == 0, head shapes other than 8 x 8): dW then regenerates the draws."""
    lib = _lib.load()
    if X.dim() != 2:
        raise ValueError(f"X: expected (N,F), got {tuple(X.shape)}")
    _chk(X, "X", contiguous=False, dtype=X.dtype)
    xcode = _dtype_code(X, "X")
    if X.stride(1) != 1:
        raise ValueError("X: rows must be contiguous")
    N, F = X.shape
    dev = X.device
    K, FP = a1.shape
    if table_dtype not in DTYPE_CODE:
        raise ValueError(f"table_dtype {table_dtype}: expected float32 or bfloat16")
    _check_heads(K, FP)
    _chk(W, "W", (F, D), device=dev)
    _chk(a1, "a1", (K, FP), device=dev)
    _chk(a2, "a2", (K, FP), device=dev)
    _chk(b1, "b1", (K,), device=dev)
    _chk(b2, "b2", (K,), device=dev)
    in_drop = _check_drop(in_drop, "in_drop")
    fts_drop = _check_drop(fts_drop, "fts_drop")
    H = torch.empty((N, D), dtype=table_dtype, device=dev)
    f1 = torch.empty((N, K), dtype=torch.float32, device=dev)
    f2 = torch.empty((N, K), dtype=torch.float32, device=dev)
    nbytes = lib.han_project_fwd_workspace(N, F, K, FP)      # > 0 only for short inputs (split-F)
    ws = _ws(nbytes, dev, "projf") if nbytes else None
    ldx = X.stride(0) if N > 1 else max(F, X.stride(0))
    keep = None
    if want_keep and in_drop > 0 and not (flags & FLAG_K1_EXACT_PIPE) and X.data_ptr() % 16 == 0:
        kb = lib.han_project_keep_bytes(N, F, ldx, K, FP)
        if kb:
            keep = torch.empty(kb, dtype=torch.uint8, device=dev)
    _lib.check(lib.han_project_fwd(
        X.data_ptr(), xcode, ldx, W.data_ptr(), a1.data_ptr(),
        a2.data_ptr(), b1.data_ptr(), b2.data_ptr(), H.data_ptr(), DTYPE_CODE[table_dtype],
        f1.data_ptr(), f2.data_ptr(), ws.data_ptr() if ws is not None else None,
        ws.numel() if ws is not None else 0, N, F, K, FP,
        in_drop, fts_drop, int(seed), _dev_word(seed_dev), int(row_offset),
        keep.data_ptr() if keep is not None else None, int(flags), _stream()), "han_project_fwd")
    if want_keep:
        return H, f1, f2, keep
    return H, f1, f2


def keep_bytes(N, F, ldx, K=8, FP=8) -> int:
    """han_project_keep_bytes: size of the keep table of an (N, F) input, 0 when the shape has none."""
    return int(_lib.load().han_project_keep_bytes(int(N), int(F), int(ldx), int(K), int(FP)))


def project_fwd_multi(X, W, a1, a2, b1, b2, in_drop=0.0, fts_drop=0.0, seeds=None, row_offset=0,
                      table_dtype=torch.float32, seed_dev=None, flags=0, want_keep=False):
    """project_fwd for all P meta-paths of ONE shared feature matrix (the reference feeds the same matrix to
    every meta-path: ex_acm3025.py:86, models/gat.py:39).  W (P,F,D), a1/a2 (P,K,F'), b1/b2 (P,K) -- the
    model's own parameter tensors.  Returns H (P,N,D), f1, f2 (P,N,K) and, with want_keep, a list of P keep
    tables (or Nones).  The eval forward of long inputs runs as ONE fused launch (X read, split and staged
    once for 4 meta-paths per block); every other case is the per-meta-path kernel, P times."""
    lib = _lib.load()
    if X.dim() != 2 or W.dim() != 3:
        raise ValueError(f"X: expected (N,F) and W (P,F,D), got {tuple(X.shape)}, {tuple(W.shape)}")
    _chk(X, "X", contiguous=False, dtype=X.dtype)
    xcode = _dtype_code(X, "X")
    if X.stride(1) != 1:
        raise ValueError("X: rows must be contiguous")
    N, F = X.shape
    dev = X.device
    P, K, FP = a1.shape
    if table_dtype not in DTYPE_CODE:
        raise ValueError(f"table_dtype {table_dtype}: expected float32 or bfloat16")
    _check_heads(K, FP)
    _chk(W, "W", (P, F, D), device=dev)
    _chk(a1, "a1", (P, K, FP), device=dev)
    _chk(a2, "a2", (P, K, FP), device=dev)
    _chk(b1, "b1", (P, K), device=dev)
    _chk(b2, "b2", (P, K), device=dev)
    in_drop = _check_drop(in_drop, "in_drop")
    fts_drop = _check_drop(fts_drop, "fts_drop")
    if (in_drop > 0 or fts_drop > 0) and (seeds is None or len(seeds) != P):
        raise ValueError("dropout needs one seed per meta-path")
    H = torch.empty((P, N, D), dtype=table_dtype, device=dev)
    f1 = torch.empty((P, N, K), dtype=torch.float32, device=dev)
    f2 = torch.empty((P, N, K), dtype=torch.float32, device=dev)
    nbytes = lib.han_project_fwd_multi_workspace(N, F, K, FP, P)
    ws = _ws(nbytes, dev, "projf") if nbytes else None
    ldx = X.stride(0) if N > 1 else max(F, X.stride(0))
    keep, kb = None, 0
    if want_keep and in_drop > 0 and not (flags & FLAG_K1_EXACT_PIPE) and X.data_ptr() % 16 == 0:
        kb = lib.han_project_keep_bytes(N, F, ldx, K, FP)
        if kb:
            keep = torch.empty((P, kb), dtype=torch.uint8, device=dev)
    seed_arr = (ctypes.c_uint64 * P)(*[int(x) & ((1 << 64) - 1) for x in (seeds if seeds is not None else [0] * P)])
    _lib.check(lib.han_project_fwd_multi(
        X.data_ptr(), xcode, ldx, W.data_ptr(), a1.data_ptr(), a2.data_ptr(), b1.data_ptr(), b2.data_ptr(),
        H.data_ptr(), DTYPE_CODE[table_dtype], f1.data_ptr(), f2.data_ptr(),
        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, N, F, K, FP, P,
        in_drop, fts_drop, seed_arr, _dev_word(seed_dev), int(row_offset),
        keep.data_ptr() if keep is not None else None, int(flags), _stream()), "han_project_fwd_multi")
    if want_keep:
        return H, f1, f2, [keep[p] if keep is not None else None for p in range(P)]
    return H, f1, f2


def project_bwd(X, dH, K, FP, in_drop=0.0, seed=0, row_offset=0, seed_dev=None, out=None, keep=None):
    """dW (F,D) = dropout_k(X)^T dH (written to `out` when given).  keep: the table project_fwd(want_keep=True)
    returned for the same X / seed (the draws are then read, not regenerated), or None."""
    lib = _lib.load()
    _chk(X, "X", contiguous=False, dtype=X.dtype)
    xcode = _dtype_code(X, "X")
    N, F = X.shape
    _chk(dH, "dH", (N, D), device=X.device)
    _check_heads(K, FP)
    dW = _out(out, "out", (F, D), X.device)
    nbytes = lib.han_project_bwd_workspace(N, F, K, FP)
    ws = _ws(nbytes, X.device, "proj")
    ldx = X.stride(0) if N > 1 else max(F, X.stride(0))
    if keep is not None:
        _chk(keep, "keep", (lib.han_project_keep_bytes(N, F, ldx, K, FP),), dtype=torch.uint8, device=X.device)
    _lib.check(lib.han_project_bwd(
        X.data_ptr(), xcode, ldx, dH.data_ptr(), dW.data_ptr(),
        ws.data_ptr(), ws.numel(), N, F, K, FP, _check_drop(in_drop, "in_drop"), int(seed), _dev_word(seed_dev),
        int(row_offset), keep.data_ptr() if keep is not None else None, _stream()), "han_project_bwd")
    return dW


def project_bwd_input(dH, W, K, FP, out=None, in_drop=0.0, seed=0, row_offset=0, seed_dev=None):
    """dX (N,F) = sum_k mask_k/keep * (dH_k @ W_k^T): gradient w.r.t. the layer input
    (only layers >= 1 of a multi-layer stack need it).  `out`: optional (N,F) view
    with unit inner stride (e.g. dM_prev[:, p, :])."""
    lib = _lib.load()
    _check_heads(K, FP)
    N = dH.shape[0]
    F = W.shape[0]
    dev = dH.device
    _chk(dH, "dH", (N, D))
    _chk(W, "W", (F, D), device=dev)
    if out is None:
        out = torch.empty((N, F), dtype=torch.float32, device=dev)
    else:
        _chk(out, "out", (N, F), device=dev, contiguous=False)
        if out.stride(1) != 1:
            raise ValueError("out: rows must be contiguous")
    _lib.check(lib.han_project_bwd_input(
        dH.data_ptr(), W.data_ptr(), out.data_ptr(), out.stride(0) if N > 1 else max(F, out.stride(0)),
        N, F, K, FP, _check_drop(in_drop, "in_drop"), int(seed), _dev_word(seed_dev), int(row_offset), _stream()),
        "han_project_bwd_input")
    return out


# --------------------------------------------------------------------------- K2
def node_attn_fwd(graph: CSRGraph, H_tab, f1, a2, b2, c, out=None, train=False, coef_drop=0.0,
                  fts_drop=0.0, seed=0, row_offset=0, activation=ACT_ELU, table_gid=None, res=None,
                  seed_dev=None, f2_src=None, f2=None):
    """utils/layers.py:26-35,46.  H_tab (NT,D): gather table of UNDROPPED projected
    rows indexed by graph.colidx (f2_j is recomputed from the gathered row with
    a2 (K,F'), b2 (K,)); with fts_drop > 0 bit 0 of each element is its keep bit
    (as project_fwd stamped it); f1 (N,K) local rows; c (D,).  table_gid (NT,)
    int32: global id of each table row when H_tab is a [local | halo] table.
    res (N,D) fp32: residual term added before the activation (layers.py:38-40).  `out`: optional (N,D) view with unit inner
    stride (e.g. M[:,p,:]).  f2_src (NT,1): one head of 64 columns only -- the neighbour scores are gathered from
    this table instead of being recomputed (slices of a head wider than 64 columns: f1 / f2_src hold the head's
    totals).  f2 (NT,K): the table rows' scores as project_fwd returned them -- with them a small graph with long rows
    (_use_lean) runs on the lean kernels, which read the scores instead of recomputing them.
    Returns out, saved where saved = (out, lse, aggp, tsum) if train else None: saved[0] is the OUTPUT view itself,
    which node_attn_bwd_rows reads (it must stay unmodified until then); the pre-activation is not stored."""
    lib = _lib.load()
    K, FP = a2.shape
    _check_heads(K, FP)
    N = graph.n_rows
    dev = H_tab.device
    _chk(H_tab, "H", (graph.n_cols, D), dtype=H_tab.dtype)
    tcode = _dtype_code(H_tab, "H")
    _chk(f1, "f1", (N, K), device=dev)
    _chk(a2, "a2", (K, FP), device=dev)
    _chk(b2, "b2", (K,), device=dev)
    _chk(c, "c", (D,), device=dev)
    if table_gid is not None:
        _chk(table_gid, "table_gid", (graph.n_cols,), dtype=torch.int32, device=dev)
    if res is not None:
        _chk(res, "res", (N, D), device=dev)
    if f2_src is not None:
        if FP != D:
            raise ValueError("f2_src: only for one head of 64 columns (K = 1, F' = 64)")
        _chk(f2_src, "f2_src", (graph.n_cols, 1), device=dev)
        f2 = f2_src
    lean = f2 is not None and table_gid is None and _use_lean(graph, H_tab)
    if lean:
        _chk(f2, "f2", (graph.n_cols, K), device=dev)
    if graph.device != dev:
        raise ValueError("graph and tables must be on the same device")
    if out is None:
        out = torch.empty((N, D), dtype=torch.float32, device=dev)
    else:
        _chk(out, "out", (N, D), device=dev, contiguous=False)
        if out.stride(1) != 1 or (N > 1 and out.stride(0) < D):
            raise ValueError("out: rows must be contiguous with row stride >= 64")
    coef_drop = _check_drop(coef_drop, "coef_drop")
    fts_drop = _check_drop(fts_drop, "fts_drop")
    if (coef_drop > 0 or fts_drop > 0) and not train:
        raise ValueError("dropout > 0 requires train=True")
    saved = None
    ptrs = [None, None, None, None]
    if train:
        # the pre-activation is not stored (round 3): the backward inverts the activation on `out`, which the
        # caller keeps anyway (it is a slice of M, the input of K3) -- 256 B per row less to write, keep and read
        aggp = torch.empty((N, D), dtype=torch.float32, device=dev)
        lse = torch.empty((N, K), dtype=torch.float32, device=dev)
        tsum = torch.empty((N, K), dtype=torch.float32, device=dev)
        saved = (out, lse, aggp, tsum)
        ptrs = [None, lse.data_ptr(), aggp.data_ptr(), tsum.data_ptr()]
    split, _keep = _row_split_arg(graph, "f", bins=not lean, split=not lean)
    dense, _keep_d = _dense_arg(graph, train, "f") if (lean and f2_src is None and _use_dense(graph, H_tab, K, FP)) else (None, None)
    timing = K2_TIMING
    if timing is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.han_node_attn_fwd(
        graph.rowptr.data_ptr(), graph.colidx.data_ptr(),
        graph.values.data_ptr() if graph.values is not None else None, H_tab.data_ptr(), tcode,
        table_gid.data_ptr() if table_gid is not None else None, f1.data_ptr(),
        f2.data_ptr() if (lean or f2_src is not None) else None,
        a2.data_ptr(), b2.data_ptr(), c.data_ptr(), res.data_ptr() if res is not None else None,
        out.data_ptr(), out.stride(0) if N > 1 else D,
        ptrs[0], ptrs[1], ptrs[2], ptrs[3], N, graph.nnz, K, FP, LEAKY_SLOPE, coef_drop, fts_drop,
        int(seed), _dev_word(seed_dev), int(row_offset), int(activation),
        (FLAG_XCD_ORDER if graph.has_locality() else 0) | (FLAG_LEAN if lean else 0) | (FLAG_K2_DEEP if K2_DEEP else 0)
        | (FLAG_K2_SHARED_HASH if K2_SHARED_HASH else 0),
        ctypes.byref(split) if split is not None else None,
        ctypes.byref(dense) if dense is not None else None, _stream()), "han_node_attn_fwd")
    if timing is not None:
        ev1.record()
        timing.append(("train" if train else "eval", ev0, ev1, N, graph.nnz))
    return out, saved


def node_attn_coefs(graph: CSRGraph, f1, f2, coef_drop=0.0, seed=0, row_offset=0, mean_heads=False,
                    table_gid=None, seed_dev=None):
    """The attention coefficients as data (attn_head(..., return_coef=True),
    utils/layers.py:27-30,43-44): (E,K) values in the CSR order of `graph`, or their
    head mean (E,) (models/gat.py:171-172).  f1 (N,K), f2 (NT,K) from project_fwd."""
    lib = _lib.load()
    _chk(f1, "f1", None)
    K = f1.shape[1]
    FP = D // K
    _check_heads(K, FP)
    dev = f1.device
    _chk(f1, "f1", (graph.n_rows, K))
    _chk(f2, "f2", (graph.n_cols, K), device=dev)
    if graph.device != dev:
        raise ValueError("graph and f1 live on different devices")
    if table_gid is not None:
        _chk(table_gid, "table_gid", (graph.n_cols,), dtype=torch.int32, device=dev)
    coef = torch.empty((graph.nnz,) if mean_heads else (graph.nnz, K), dtype=torch.float32, device=dev)
    _lib.check(lib.han_node_attn_coefs(
        graph.rowptr.data_ptr(), graph.colidx.data_ptr(),
        graph.values.data_ptr() if graph.values is not None else None,
        table_gid.data_ptr() if table_gid is not None else None, f1.data_ptr(), f2.data_ptr(),
        coef.data_ptr(), int(bool(mean_heads)), graph.n_rows, graph.nnz, K, FP, LEAKY_SLOPE,
        _check_drop(coef_drop, "coef_drop"), int(seed), _dev_word(seed_dev), int(row_offset), _stream()), "han_node_attn_coefs")
    return coef


def gs_row_bytes(K=8, FP=8, table_dtype=torch.float32) -> int:
    """Bytes of one fused backward row [g | (f1, lse, s, 0) x K] (han_gs_row_bytes)."""
    n = int(_lib.load().han_gs_row_bytes(K, FP, DTYPE_CODE[table_dtype]))
    if n == 0:
        raise NotImplementedError(f"no fused backward row for K={K}, FP={FP}, {table_dtype}")
    return n


def gs_views(gs, K=8, FP=8, table_dtype=torch.float32):
    """(g (N,D) table_dtype, stats (N,K,4) fp32) views of a fused backward table (N, row_bytes) uint8."""
    gb = D * (2 if table_dtype == torch.bfloat16 else 4)
    g = gs[:, :gb].view(table_dtype)
    stats = gs[:, gb:gb + 16 * K].view(torch.float32).unflatten(1, (K, 4))
    return g, stats


def node_attn_bwd_rows(dOut, out, aggp, tsum, f1, lse, c, activation=ACT_ELU, K=8, FP=8,
                       table_dtype=torch.float32, res=None, dc_out=None, gs_out=None):
    """Row-local half of the K2 backward.  dOut (N,D) view (unit inner stride); out (N,D) view: the forward's
    output rows (saved[0] of node_attn_fwd), from which the pre-activation is recovered.
    Returns gs (N, row_bytes) uint8 -- the fused table [g | (f1, lse, s, 0) x K] the transposed-graph
    pass gathers from (gs_views() splits it) --, df1 (N,K), dc (D,) [written to dc_out when given].
    gs_out: optional preallocated destination (e.g. the local block of an exchange table)."""
    lib = _lib.load()
    _check_heads(K, FP)
    N = out.shape[0]
    dev = out.device
    _chk(dOut, "dOut", (N, D), device=dev, contiguous=False)
    _chk(out, "out", (N, D), device=dev, contiguous=False)
    if dOut.stride(1) != 1 or out.stride(1) != 1:
        raise ValueError("dOut / out: rows must be contiguous")
    for t, n, s in ((aggp, "aggp", (N, D)), (tsum, "tsum", (N, K)),
                    (f1, "f1", (N, K)), (lse, "lse", (N, K)), (c, "c", (D,))):
        _chk(t, n, s, device=dev)
    rb = gs_row_bytes(K, FP, table_dtype)
    if gs_out is None:      # rows padded to whole lines: keep the padding bytes defined
        padded = rb != D * (2 if table_dtype == torch.bfloat16 else 4) + 16 * K
        gs = (torch.zeros if padded else torch.empty)((N, rb), dtype=torch.uint8, device=dev)
    else:
        gs = _chk(gs_out, "gs_out", (N, rb), dtype=torch.uint8, device=dev)
    df1 = torch.empty((N, K), dtype=torch.float32, device=dev)
    dc = _out(dc_out, "dc_out", (D,), dev)
    ws = _ws(lib.han_node_attn_bwd_workspace(N, K, FP), dev, "rows")
    _lib.check(lib.han_node_attn_bwd_rows(
        dOut.data_ptr(), dOut.stride(0) if N > 1 else D, out.data_ptr(), out.stride(0) if N > 1 else D, aggp.data_ptr(),
        tsum.data_ptr(), f1.data_ptr(), lse.data_ptr(), c.data_ptr(),
        res.data_ptr() if res is not None else None, gs.data_ptr(),
        DTYPE_CODE[table_dtype], df1.data_ptr(), dc.data_ptr(), ws.data_ptr(), ws.numel(), N, K, FP,
        int(activation), _stream()), "han_node_attn_bwd_rows")
    return gs, df1, dc


def node_attn_bwd_cols(graph_t: CSRGraph, gs_tab, H, f2, df1, a1, a2, coef_drop=0.0,
                       fts_drop=0.0, seed=0, src_offset=0, dst_offset=0, table_gid=None, seed_dev=None):
    """Transposed-graph half of the K2 backward.  graph_t rows = local sources j,
    its colidx = destinations i indexing the fused table gs_tab (NT, row_bytes) uint8.
    H (NS,D) undropped local rows (keep bits in bit 0 when fts_drop > 0),
    f2/df1 (NS,K).  Returns dH (NS,D), df2 (NS,K)."""
    lib = _lib.load()
    K, FP = a1.shape
    _check_heads(K, FP)
    NS = graph_t.n_rows
    dev = H.device
    _chk(H, "H", (NS, D), dtype=H.dtype)
    tcode = _dtype_code(H, "H")
    _chk(gs_tab, "gs", (graph_t.n_cols, gs_row_bytes(K, FP, H.dtype)), device=dev, dtype=torch.uint8)
    if table_gid is not None:
        _chk(table_gid, "table_gid", (graph_t.n_cols,), dtype=torch.int32, device=dev)
    _chk(f2, "f2", (NS, K), device=dev)
    _chk(df1, "df1", (NS, K), device=dev)
    _chk(a1, "a1", (K, FP), device=dev)
    _chk(a2, "a2", (K, FP), device=dev)
    fts_drop = _check_drop(fts_drop, "fts_drop")
    dH = torch.empty((NS, D), dtype=torch.float32, device=dev)
    df2 = torch.empty((NS, K), dtype=torch.float32, device=dev)
    lean_b = not graph_t.masked and _use_lean(graph_t, H)
    split, _keep = _row_split_arg(graph_t, "b", bins=not lean_b, split=not lean_b)
    dense, _keep_d = _dense_arg(graph_t, False, "b") if (lean_b and table_gid is None and _use_dense(graph_t, H, K, FP)) else (None, None)
    timing = K2_TIMING
    if timing is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.han_node_attn_bwd_cols(
        graph_t.rowptr.data_ptr(), graph_t.colidx.data_ptr(),
        graph_t.values.data_ptr() if graph_t.values is not None else None, gs_tab.data_ptr(),
        table_gid.data_ptr() if table_gid is not None else None, H.data_ptr(),
        tcode, f2.data_ptr(), df1.data_ptr(), a1.data_ptr(), a2.data_ptr(), dH.data_ptr(), df2.data_ptr(),
        NS, graph_t.nnz, K, FP, LEAKY_SLOPE, _check_drop(coef_drop, "coef_drop"), fts_drop,
        int(seed), _dev_word(seed_dev), int(src_offset), int(dst_offset),
        (FLAG_XCD_ORDER if graph_t.has_locality() else 0) | (FLAG_MASKED_EDGES if graph_t.masked else 0)
        | (FLAG_LEAN if lean_b else 0),
        ctypes.byref(split) if split is not None else None,
        ctypes.byref(dense) if dense is not None else None, _stream()), "han_node_attn_bwd_cols")
    if timing is not None:
        ev1.record()
        timing.append(("bwd_cols", ev0, ev1, NS, graph_t.nnz))
    return dH, df2


def score_param_bwd(H, df1, df2, K=8, FP=8, out=None):
    """da1, da2 (K,F'), db1, db2 (K,) [written to the 4 tensors of `out` when given]."""
    lib = _lib.load()
    _check_heads(K, FP)
    N = H.shape[0]
    dev = H.device
    _chk(H, "H", (N, D), dtype=H.dtype)
    tcode = _dtype_code(H, "H")
    _chk(df1, "df1", (N, K), device=dev)
    _chk(df2, "df2", (N, K), device=dev)
    o = out if out is not None else (None,) * 4
    da1 = _out(o[0], "da1", (K, FP), dev)
    da2 = _out(o[1], "da2", (K, FP), dev)
    db1 = _out(o[2], "db1", (K,), dev)
    db2 = _out(o[3], "db2", (K,), dev)
    ws = _ws(lib.han_score_param_bwd_workspace(N, K, FP), dev, "score")
    _lib.check(lib.han_score_param_bwd(
        H.data_ptr(), tcode, df1.data_ptr(), df2.data_ptr(), da1.data_ptr(), da2.data_ptr(),
        db1.data_ptr(), db2.data_ptr(), ws.data_ptr(), ws.numel(), N, K, FP, _stream()),
        "han_score_param_bwd")
    return da1, da2, db1, db2


# --------------------------------------------------------------------------- K3
def sem_attn_fwd(M, w_omega, b_omega, u_omega, flags=0):
    """utils/layers.py:152-159.  M (N,P,D) -> Z (N,D), beta (N,P)."""
    lib = _lib.load()
    _chk(M, "M")
    if M.dim() != 3 or M.shape[2] % 64 != 0 or M.shape[2] == 0:
        raise ValueError(f"M: expected (N,P,D) with D a multiple of 64, got {tuple(M.shape)}")
    N, P, Dm = M.shape
    A = w_omega.shape[1]
    dev = M.device
    _chk(w_omega, "w_omega", (Dm, A), device=dev)
    _chk(b_omega, "b_omega", (A,), device=dev)
    _chk(u_omega, "u_omega", (A,), device=dev)
    Z = torch.empty((N, Dm), dtype=torch.float32, device=dev)
    beta = torch.empty((N, P), dtype=torch.float32, device=dev)
    _lib.check(lib.han_sem_attn_fwd(M.data_ptr(), w_omega.data_ptr(), b_omega.data_ptr(),
                                    u_omega.data_ptr(), Z.data_ptr(), beta.data_ptr(), N, P, Dm, A,
                                    int(flags), _stream()), "han_sem_attn_fwd")
    return Z, beta


def sem_attn_bwd(M, w_omega, b_omega, u_omega, beta, dZ, out=None, flags=0):
    """dM, dw, db, du [the last three written to the tensors of `out` when given]."""
    lib = _lib.load()
    N, P, Dm = M.shape
    A = w_omega.shape[1]
    dev = M.device
    _chk(M, "M", (N, P, Dm))
    _chk(beta, "beta", (N, P), device=dev)
    _chk(dZ, "dZ", (N, Dm), device=dev)
    dM = torch.empty_like(M)
    o = out if out is not None else (None,) * 3
    dw = _out(o[0], "dw", tuple(w_omega.shape), dev)
    db = _out(o[1], "db", tuple(b_omega.shape), dev)
    du = _out(o[2], "du", tuple(u_omega.shape), dev)
    ws = _ws(lib.han_sem_attn_bwd_workspace(N, P, Dm, A), dev, "sem")
    _lib.check(lib.han_sem_attn_bwd(
        M.data_ptr(), w_omega.data_ptr(), b_omega.data_ptr(), u_omega.data_ptr(), beta.data_ptr(),
        dZ.data_ptr(), dM.data_ptr(), dw.data_ptr(), db.data_ptr(), du.data_ptr(), ws.data_ptr(),
        ws.numel(), N, P, Dm, A, int(flags), _stream()), "han_sem_attn_bwd")
    return dM, dw, db, du


# ------------------------------------------------------------- classifier + loss
def classifier_loss(Z, Wc, bc, labels, mask, row_weight, backward=False, grad_out=None):
    """models/gat.py:65-72 + models/base_gattn.py:41-48,61-69.
    Z (N,D); Wc (HC,D,C); bc (HC,C); labels int32 (N,); mask uint8 (N,).
    Returns logits (N,C), loss_acc (2,) [masked CE, masked accuracy] and, if
    backward, (dZ, dWc, dbc) [dWc, dbc written to the tensors of grad_out when given]."""
    lib = _lib.load()
    _chk(Z, "Z")
    N = Z.shape[0]
    dev = Z.device
    HC, Dm, C = Wc.shape
    if Dm % 64 != 0 or Dm == 0:
        raise ValueError(f"Wc: the embedding width must be a multiple of 64 (zero-pad it), got {Dm}")
    _chk(Z, "Z", (N, Dm))
    _chk(Wc, "Wc", (HC, Dm, C), device=dev)
    _chk(bc, "bc", (HC, C), device=dev)
    _chk(labels, "labels", (N,), dtype=torch.int32, device=dev)
    _chk(mask, "mask", (N,), dtype=torch.uint8, device=dev)
    logits = torch.empty((N, C), dtype=torch.float32, device=dev)
    loss_acc = torch.empty((2,), dtype=torch.float32, device=dev)
    grads = None
    ptrs = (None, None, None)
    if backward:
        dZ = torch.empty((N, Dm), dtype=torch.float32, device=dev)
        go = grad_out if grad_out is not None else (None, None)
        dWc = _out(go[0], "dWc", tuple(Wc.shape), dev)
        dbc = _out(go[1], "dbc", tuple(bc.shape), dev)
        grads = (dZ, dWc, dbc)
        ptrs = (dZ.data_ptr(), dWc.data_ptr(), dbc.data_ptr())
    ws = _ws(lib.han_classifier_workspace(N, Dm, C, HC), dev, "cls")
    _lib.check(lib.han_classifier_loss(
        Z.data_ptr(), Wc.data_ptr(), bc.data_ptr(), labels.data_ptr(), mask.data_ptr(),
        float(row_weight), logits.data_ptr(), loss_acc.data_ptr(), ptrs[0], ptrs[1], ptrs[2],
        ws.data_ptr(), ws.numel(), N, Dm, C, HC, _stream()), "han_classifier_loss")
    return logits, loss_acc, grads


def classifier_bwd(Z, Wc, bc, dlogits):
    """Backward of logits = (1/HC) sum_h (Z Wc[h] + bc[h]) (models/gat.py:65-72) for a given dlogits (N,C):
    returns (dZ, dWc, dbc)."""
    lib = _lib.load()
    _chk(Z, "Z")
    N, dev = Z.shape[0], Z.device
    HC, Dm, C = Wc.shape
    if Dm % 64 != 0 or Dm == 0:
        raise ValueError(f"Wc: the embedding width must be a multiple of 64 (zero-pad it), got {Dm}")
    _chk(Z, "Z", (N, Dm))
    _chk(Wc, "Wc", (HC, Dm, C), device=dev)
    _chk(bc, "bc", (HC, C), device=dev)
    _chk(dlogits, "dlogits", (N, C), device=dev)
    dZ = torch.empty((N, Dm), dtype=torch.float32, device=dev)
    dWc, dbc = torch.empty_like(Wc), torch.empty_like(bc)
    ws = _ws(lib.han_classifier_bwd_workspace(N, Dm, C, HC), dev, "clsb")
    _lib.check(lib.han_classifier_bwd(Z.data_ptr(), Wc.data_ptr(), bc.data_ptr(), dlogits.data_ptr(), dZ.data_ptr(),
                                      dWc.data_ptr(), dbc.data_ptr(), ws.data_ptr(), ws.numel(), N, Dm, C, HC,
                                      _stream()), "han_classifier_bwd")
    return dZ, dWc, dbc


# ------------------------------------------------------------------- optimiser
def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, l2_coef=0.0, step_dev=None):
    """lr_t = lr*sqrt(1-b2^t)/(1-b1^t) from the caller, or -- with step_dev (device step
    count t) -- the base rate lr, the correction then being computed on the device."""
    lib = _lib.load()
    n = param.numel()
    for t, nme in ((param, "param"), (grad, "grad"), (m, "m"), (v, "v")):
        _chk(t, nme, (n,), device=param.device)
    _lib.check(lib.han_adam_step(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n,
                                 float(lr_t), float(beta1), float(beta2), float(eps),
                                 float(l2_coef), _dev_word(step_dev), _stream()), "han_adam_step")


def l2_half_sumsq(param):
    lib = _lib.load()
    _chk(param, "param", (param.numel(),))
    out = torch.empty((1,), dtype=torch.float32, device=param.device)
    ws = _ws(4096 * 4, param.device, "l2")
    _lib.check(lib.han_l2_half_sumsq(param.data_ptr(), param.numel(), out.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream()), "han_l2_half_sumsq")
    return out
