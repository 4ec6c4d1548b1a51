"""Layer library: the reference's ``utils/layers.py`` surface on the HIP kernels.

Same names, argument meaning and return values as the reference:

* :func:`attn_head`       -- utils/layers.py:7-46  (dense additive-mask form)
* :func:`attn_head_const_1` -- utils/layers.py:49-81 (HAN_nd ablation)
* :func:`sp_attn_head`    -- utils/layers.py:85-127 (SparseTensor form)
* :func:`SimpleAttLayer`  -- utils/layers.py:132-164

TensorFlow creates the variables inside each call; here they are passed in
``params`` (functional form) or owned by ``han_amd.gat.HeteGAT_multi`` (module
form).  The fast path used by the model is :class:`NodeLevelAttention`, which
runs all K heads of all P meta-paths through the K1/K2 kernels and writes the
heads straight into ``M[:, p, :]`` (models/gat.py:46,58-60).

There is no CPU implementation: every function raises on non-GPU tensors.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F_torch

from . import ops, rng
from .dist import NodePartition
from .graph import as_graph

D = ops.D


def _act_code(activation):
    """Map the reference's `activation` argument to (kernel code, torch post-op)."""
    if activation is None:
        return ops.ACT_IDENTITY, None
    if activation in (F_torch.elu, torch.nn.functional.elu, "elu") or \
            isinstance(activation, torch.nn.ELU):
        return ops.ACT_ELU, None
    if activation == "identity":
        return ops.ACT_IDENTITY, None
    if callable(activation):
        return ops.ACT_IDENTITY, activation   # kernel emits the pre-activation; torch applies it
    raise ValueError(f"unsupported activation {activation!r}")


def _direct(p) -> bool:
    """True when parameter `p` asks the backward kernels to write straight into p.grad
    (set by HeteGAT_multi.direct_grads(True); see NodeLevelAttention.forward)."""
    return bool(getattr(p, "_han_direct_grad", False)) and p.grad is not None


def _same_tensor(ts) -> bool:
    """True when every entry is the same view of the same storage (the meta-paths share their features)."""
    t0 = ts[0]
    return all(t.data_ptr() == t0.data_ptr() and t.shape == t0.shape and t.stride() == t0.stride()
               and t.dtype == t0.dtype for t in ts[1:])


class _on_path:
    """Run the enclosed launches of meta-path p on its own stream (cfg["streams"], single GPU): the per-meta-path
    chains K1 -> K2 (forward) and rows -> cols -> score gradients -> dW (backward) are independent of each other,
    and at the size of the reference's data sets every kernel is a few microseconds on a few CUs -- side by side
    in a captured epoch they overlap instead of queueing (HANTrainer(use_graph=True)).  Scratch buffers are per path."""

    def __init__(self, streams, p):
        self.s = streams[p] if streams is not None else None
        self.p = p

    def __enter__(self):
        if self.s is not None:
            self.ctx = torch.cuda.stream(self.s)
            self.ctx.__enter__()
            self.prev, ops.WS_SUFFIX = ops.WS_SUFFIX, ops.WS_SUFFIX + f"@p{self.p}"

    def __exit__(self, *exc):
        if self.s is not None:
            ops.WS_SUFFIX = self.prev
            self.ctx.__exit__(*exc)


class _on_side:
    """Run the enclosed launches on the side stream (cfg["side_stream"]: single GPU, eager training step on large
    graphs).  The backward's dW of meta-path p needs only dH_p, so it runs beside the transposed-graph gather of
    meta-path p + 1: the gather is bound by the memory fabric and leaves vector / matrix issue slots that dW, bound by
    exactly those, can use -- 2.70 + 0.52 ms one after the other, 2.95 ms side by side.  What was measured and is NOT
    done: the forward's K1 of meta-path p + 1 beside K2 of meta-path p gains nothing (K2's blocks refill every slot
    they free, K1's 8-wave blocks with 40 KB of LDS wait: 2.5 ms instead of 0.74, a higher stream priority changes
    nothing, and CU masks only move the same CU time around: tools/cu_mask_probe.py); two K2 launches beside each
    other evict each other's gather table from the Infinity Cache (four meta-path chains side by side: 40.7 ms per
    epoch against 35.5).  Scratch buffers of the side stream are its own (ops.WS_SUFFIX)."""

    def __init__(self, side):
        self.s = side

    def __enter__(self):
        self.ctx = torch.cuda.stream(self.s)
        self.ctx.__enter__()
        self.prev, ops.WS_SUFFIX = ops.WS_SUFFIX, ops.WS_SUFFIX + "@side"

    def __exit__(self, *exc):
        ops.WS_SUFFIX = self.prev
        self.ctx.__exit__(*exc)


def _used_on(stream, *tensors):
    """Tensors allocated on one stream and read on another: tell the caching allocator, so that their memory is not
    handed out again before that stream is done with it."""
    for t in tensors:
        if t is not None:
            t.record_stream(stream)


def _fork(streams):
    if streams is not None:
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)


def _path_order(streams, graphs, how):
    """Order in which the per-meta-path chains are issued on their own streams (a captured epoch).  The chains are
    independent, so the results do not depend on it; what does is how the runtime lays the graph's branches on its
    queues (the first child of a fork continues on the parent's queue, the others start on queues of their own).
    "heavy" = largest graph first: measured with HANTrainer(overlap_eval="branch"), where the training forward's
    second chain otherwise starts ~50 us late (DBLP-like 0.88 -> 0.83 ms per epoch, ACM-like unchanged); the plain
    captured epoch keeps 0 .. P-1 (heavy-first there: ACM-like 0.394 -> 0.384 ms but DBLP-like 0.859 -> 0.902)."""
    P = len(graphs)
    if streams is None or how != "heavy":
        return list(range(P))
    return sorted(range(P), key=lambda p: (-int(graphs[p].nnz), p))


def _join(streams):
    if streams is not None:
        cur = torch.cuda.current_stream()
        for st in streams:
            cur.wait_stream(st)


class _Ready:
    """A table that needs no exchange (same interface as the async exchange handles)."""

    def __init__(self, table):
        self.table = table

    def wait(self):
        return self.table


class NodeLevelAttention(torch.autograd.Function):
    """K1 + K2 for every meta-path: (X_p, graph_p) -> M (N, P, D).

    forward(Xin, W (P,F,D), a1 (P,K,F'), b1 (P,K), a2, b2, c (P,D), Wr, br, xs, graphs, cfg)
      Wr, br  None, or the residual connection of utils/layers.py:38-40 for layers whose
              input width differs from the head width: Wr (P,F,D) = the K heads'
              conv1d(seq, F', 1) kernels side by side, br (P,D) their biases; the term
              dropout_k(X) @ Wr_k + br_k is added before the activation
      Xin     None for the first layer; for layers >= 1 (models/gat.py:48-57) the
              previous layer's output (N,P,F): meta-path p reads Xin[:, p, :] and the
              backward returns dXin
      xs      tuple of P feature tensors (N,F) (no gradient: they are inputs);
              ignored when Xin is given
      graphs  tuple of P CSRGraph (rows = local destinations)
      cfg     dict: train (bool), in_drop, coef_drop, seeds (tuple of P ints),
              act (kernel activation code), part (NodePartition or None),
              table_dtype (torch.float32 | torch.bfloat16: storage of the H / g tables),
              graphs_t (tuple of P transposed graphs for the backward, or None),
              plans_f / plans_b (per meta-path HaloPlan or None: halo exchange instead of
              the all-gather, forward / backward tables),
              coef_sink (None, or a list that receives per meta-path the coefficients
              of this call as data -- (E,K), or their head mean (E,) with coef_mean=True;
              return_coef of layers.py:43-44 / models/gat.py:143-172; single GPU only)
    """

    @staticmethod
    def forward(ctx, Xin, W, a1, b1, a2, b2, c, Wr, br, xs, graphs, cfg):
        P = len(graphs)
        if Xin is not None:
            Xin = Xin.contiguous()
            xs = tuple(Xin[:, p, :] for p in range(P))
        K, FP = a1.shape[1], a1.shape[2]
        part: NodePartition | None = cfg.get("part")
        train = bool(cfg["train"])
        in_drop = float(cfg.get("in_drop", 0.0)) if train else 0.0
        coef_drop = float(cfg.get("coef_drop", 0.0)) if train else 0.0
        N = xs[0].shape[0]
        M = torch.empty((N, P, D), dtype=torch.float32, device=W.device)
        row_offset = part.row_start if part is not None else 0
        saved = [None] * P
        seed_dev = cfg.get("seed_dev")      # device seed word of a captured step (see han_hip.h "Seeds")
        multi = part is not None and part.active
        plans_f = cfg.get("plans_f") if multi else None      # per meta-path HaloPlan or None
        # all projections first, each table's all-gather started as soon as it exists:
        # the exchange of meta-path p+1.. overlaps the node attention of meta-path p
        proj, proj_keep = [None] * P, [None] * P
        xs_full = cfg.get("xs_full") if (multi and Xin is None) else None
        tdt = cfg.get("table_dtype", torch.float32)
        # the reference feeds ONE feature matrix to every meta-path (ex_acm3025.py:86): all P projections then go
        # through ONE call -- the eval forward of long inputs as one fused launch that reads, splits and stages
        # every X tile once for four meta-paths (ops.project_fwd_multi)
        replicated = [xs_full is not None and (plans_f is None or plans_f[p] is None) for p in range(P)]
        pj = [None] * P
        src = xs_full if all(replicated) else (xs if not any(replicated) else None)
        streams = cfg.get("streams") if (not multi and cfg.get("streams") is not None and len(cfg["streams"]) >= P) else None
        # the backward runs dW beside the next meta-path's gather (_on_side); the forward stays one chain
        side = cfg.get("side_stream") if (streams is None and not multi and train and P > 1 and W.is_cuda) else None
        _fork(streams)
        # HANTrainer(overlap_eval=True): the eval forward's K1 + K2 as one more branch of this fork / join section
        overlap = cfg.get("overlap") if (train and cfg.get("group", 0) == 0) else None
        if overlap is not None:
            overlap.node_level()
        if streams is None and P > 1 and src is not None and _same_tensor(src) and W.is_contiguous() and src[0].stride(-1) == 1:
            full = all(replicated)
            Hs, f1s, f2s, keeps = ops.project_fwd_multi(src[0], W, a1, a2, b1, b2, in_drop=in_drop, fts_drop=in_drop,
                                                        seeds=[int(v) for v in cfg["seeds"]],
                                                        row_offset=0 if full else row_offset, table_dtype=tdt,
                                                        seed_dev=seed_dev, want_keep=True)
            pj = [(Hs[p], f1s[p], f2s[p], keeps[p]) for p in range(P)]

        def project_path(p):
            seed = int(cfg["seeds"][p])
            plan = plans_f[p] if plans_f is not None else None
            handle = keep = None
            if replicated[p]:
                # replicated projection: every rank holds the features of ALL rows and projects the whole
                # table itself instead of receiving (G-1)/G of it -- a point-to-point xGMI link moves a
                # 256-B row slower than K1 recomputes it (dist.replication_policy).  Masks are keyed by
                # global row ids, so the rows are bit-identical to what their owners compute.
                Hf, f1f, f2f, keepf = pj[p] if pj[p] is not None else ops.project_fwd(
                    xs_full[p], W[p], a1[p], a2[p], b1[p], b2[p], in_drop=in_drop, fts_drop=in_drop, seed=seed,
                    row_offset=0, table_dtype=tdt, seed_dev=seed_dev, want_keep=True)
                r0, r1 = part.row_start, part.row_end
                H, f1, f2 = Hf[r0:r1], f1f[r0:r1], f2f[r0:r1]
                if keepf is not None:      # the local rows of the table (+ its slack) when they form a table themselves
                    Fw = xs_full[p].shape[1]
                    kb_loc = ops.keep_bytes(r1 - r0, Fw, xs[p].stride(0), K, FP)
                    keep = keepf[r0 * Fw:r0 * Fw + kb_loc] if kb_loc and (r0 * Fw) % 8 == 0 else None
                handle = _Ready(Hf)
            else:
                # training: the forward also writes the keep table of its per-head input dropout, which dW
                # reads instead of regenerating the draws (None for shapes without a table)
                H, f1, f2, keep = pj[p] if pj[p] is not None else ops.project_fwd(
                    xs[p], W[p], a1[p], a2[p], b1[p], b2[p], in_drop=in_drop, fts_drop=in_drop, seed=seed,
                    row_offset=row_offset, table_dtype=tdt, seed_dev=seed_dev, want_keep=True)
            if multi and handle is None:      # halo rows only (HaloPlan) or the whole shard (all-gather)
                tag = ("f", cfg.get("layer", 0), cfg.get("group", 0), p)   # persistent exchange table of this (layer, head group, meta-path)
                handle = plan.exchange_async(H, tag) if plan is not None else part.all_gather_rows_async(H, tag)
            if cfg.get("coef_sink") is not None:
                if multi:
                    raise NotImplementedError("return_coef is not provided under a node partition")
                cfg["coef_sink"].append(ops.node_attn_coefs(
                    graphs[p], f1, f2, coef_drop=coef_drop, seed=seed, row_offset=row_offset,
                    mean_heads=bool(cfg.get("coef_mean", False)), seed_dev=seed_dev))
            R = None
            if Wr is not None:   # same seed -> the same per-head input-dropout draws as for H
                R, _, _ = ops.project_fwd(xs[p], Wr[p], a1[p], a2[p], b1[p], b2[p], in_drop=in_drop,
                                          fts_drop=0.0, seed=seed, row_offset=row_offset, seed_dev=seed_dev)
                R = R + br[p]
            proj[p] = (H, f1, f2, handle, R)
            proj_keep[p] = keep

        def attend_path(p):
            H, f1, f2, handle, R = proj[p]
            H_tab = handle.wait() if multi else H
            plan = plans_f[p] if plans_f is not None else None
            _, sv = ops.node_attn_fwd(plan.graph if plan is not None else graphs[p], H_tab, f1, a2[p], b2[p],
                                      c[p], out=M[:, p, :], train=train, coef_drop=coef_drop,
                                      fts_drop=in_drop, seed=int(cfg["seeds"][p]), row_offset=row_offset,
                                      activation=cfg["act"],
                                      table_gid=plan.gid if plan is not None else None, res=R,
                                      seed_dev=seed_dev, f2=None if multi else f2)
            if train:
                # sv[0] is the OUTPUT view M[:, p, :] the backward inverts the activation on: it is kept through
                # ctx.save_for_backward(M) below (version-checked by autograd, no reference cycle), not here
                saved[p] = (H, f1, f2, None) + sv[1:] + (R, proj_keep[p])

        order = _path_order(streams, graphs, cfg.get("path_order"))
        for p in order:
            with _on_path(streams, p):
                project_path(p)
        for p in order:
            with _on_path(streams, p):
                attend_path(p)
        _join(streams)
        if overlap is not None:
            overlap.join()
        ctx.overlap = overlap
        del proj
        ctx.side = side
        ctx.cfg, ctx.xs, ctx.graphs = cfg, xs, graphs
        ctx.xin_shape = tuple(Xin.shape) if Xin is not None else None
        ctx.saved_per_p = saved
        # the per-meta-path tensors above were allocated on these streams: the backward must run each meta-path on
        # the SAME stream (the caching allocator ties a block to its allocation stream), whatever cfg holds by then
        ctx.streams = streams
        ctx.in_drop, ctx.coef_drop = in_drop, coef_drop
        ctx.has_res = Wr is not None
        # direct-gradient mode (HANTrainer): every parameter carries a pre-bound .grad slice of
        # the flat gradient buffer and is used once per step, so the backward kernels WRITE
        # their results there and autograd gets None (no copy / accumulate launches)
        plist = (W, a1, b1, a2, b2, c) + ((Wr, br) if Wr is not None else ())
        ctx.direct = tuple(p.grad for p in plist) if all(_direct(p) for p in plist) else None
        # M, the output, is backward state too (the pre-activation is recovered from it): saved through autograd, so
        # that an in-place change of M between forward and backward raises instead of giving wrong gradients
        ctx.save_for_backward(W, a1, b1, a2, b2, c, *((Wr,) if Wr is not None else ()), *((M,) if train else ()))
        return M

    @staticmethod
    def backward(ctx, dM):
        W, a1, b1, a2, b2, c = ctx.saved_tensors[:6]
        Wr = ctx.saved_tensors[6] if ctx.has_res else None
        if not ctx.cfg["train"]:
            raise RuntimeError("NodeLevelAttention was run with train=False; no backward state")
        Mout = ctx.saved_tensors[-1]
        dWr = torch.empty_like(Wr) if Wr is not None else None
        dbr = torch.empty_like(c) if Wr is not None else None
        cfg, xs, graphs = ctx.cfg, ctx.xs, ctx.graphs
        if not cfg["train"]:
            raise RuntimeError("NodeLevelAttention was run with train=False; no backward state")
        part: NodePartition | None = cfg.get("part")
        P = len(graphs)
        K, FP = a1.shape[1], a1.shape[2]
        dM = dM.contiguous()
        graphs_t = cfg.get("graphs_t") or tuple(g.transpose() for g in graphs)
        row_offset = part.row_start if part is not None else 0
        direct = ctx.direct
        if direct is not None:
            dW, da1, db1, da2, db2, dc = direct[:6]
            dWr, dbr = (direct[6], direct[7]) if Wr is not None else (None, None)
        else:
            dW = torch.empty_like(W)
            da1, da2 = torch.empty_like(a1), torch.empty_like(a2)
            db1, db2 = torch.empty_like(b1), torch.empty_like(b2)
            dc = torch.empty_like(c)
        dXin = None
        if ctx.xin_shape is not None and ctx.needs_input_grad[0]:
            dXin = torch.empty(ctx.xin_shape, dtype=torch.float32, device=W.device)
        multi = part is not None and part.active
        seed_dev = cfg.get("seed_dev")
        plans_b = cfg.get("plans_b") if multi else None
        masked = cfg.get("masked_bwd")        # per meta-path MaskedBackwardPlan, or None (the full pass)
        rows = [None] * P
        dres_in = []
        streams = ctx.streams       # the streams the forward ran (and allocated) on
        side = ctx.side
        _fork(streams)
        if ctx.overlap is not None:      # the eval forward's K3 + classifier beside the per-meta-path backward chains
            ctx.overlap.head()

        def rows_path(p):      # row-local halves first; their tables go out while we continue
            H, f1, f2, _, lse, aggp, tsum, R, _keep = ctx.saved_per_p[p]
            gs, df1, dcp = ops.node_attn_bwd_rows(dM[:, p, :], Mout[:, p, :], aggp, tsum, f1, lse, c[p],
                                                  activation=cfg["act"], K=K, FP=FP,
                                                  table_dtype=H.dtype, res=R, dc_out=dc[p])
            if Wr is not None:      # residual: d(pre) = g flows into Wr, br and the input
                g32 = ops.gs_views(gs, K, FP, H.dtype)[0].to(torch.float32).contiguous()
                seed_p = int(cfg["seeds"][p])
                dbr[p] = dcp
                ops.project_bwd(xs[p], g32, K, FP, in_drop=ctx.in_drop, seed=seed_p,
                                row_offset=row_offset, seed_dev=seed_dev, out=dWr[p])
                if dXin is not None:
                    dres_in.append(ops.project_bwd_input(g32, Wr[p], K, FP, in_drop=ctx.in_drop,
                                                         seed=seed_p, row_offset=row_offset,
                                                         seed_dev=seed_dev))
            mb = masked[p] if masked is not None else None
            if mb is not None:        # opt-in masked backward: only the live rows of [g | stats] are read / travel
                rows[p] = (mb.table_async(gs, ("bm", cfg.get("layer", 0), cfg.get("group", 0), p)), df1)
            elif multi:
                plan = plans_b[p] if plans_b is not None else None
                ex = plan.exchange_async if plan is not None else part.all_gather_rows_async
                rows[p] = (ex(gs, ("b", cfg.get("layer", 0), cfg.get("group", 0), p)), df1)   # ONE fused [g | stats] table on the wire
            else:
                rows[p] = (gs, df1)

        def cols_path(p):
            H, f1, f2, pre, lse, aggp, tsum, R, keep = ctx.saved_per_p[p]
            seed = int(cfg["seeds"][p])
            gs_h, df1 = rows[p]
            mb = masked[p] if masked is not None else None
            gs_tab = gs_h.wait() if (multi or mb is not None) else gs_h
            plan = plans_b[p] if (plans_b is not None and mb is None) else None
            gt = mb.graph_t if mb is not None else (plan.graph if plan is not None else graphs_t[p])
            dH, df2 = ops.node_attn_bwd_cols(gt, gs_tab, H, f2, df1, a1[p], a2[p],
                                             coef_drop=ctx.coef_drop, fts_drop=ctx.in_drop, seed=seed,
                                             src_offset=row_offset, dst_offset=0,
                                             table_gid=mb.gid if mb is not None else (plan.gid if plan is not None else None),
                                             seed_dev=seed_dev)
            rows[p] = None
            ops.score_param_bwd(H, df1, df2, K=K, FP=FP, out=(da1[p], da2[p], db1[p], db2[p]))
            if side is not None and dXin is None:      # dW(p) beside the transposed-graph gather of meta-path p + 1
                side.wait_stream(torch.cuda.current_stream())
                with _on_side(side):
                    ops.project_bwd(xs[p], dH, K, FP, in_drop=ctx.in_drop, seed=seed,
                                    row_offset=row_offset, seed_dev=seed_dev, out=dW[p], keep=keep)
                _used_on(side, dH)
                return
            ops.project_bwd(xs[p], dH, K, FP, in_drop=ctx.in_drop, seed=seed,
                            row_offset=row_offset, seed_dev=seed_dev, out=dW[p], keep=keep)
            if dXin is not None:
                ops.project_bwd_input(dH, W[p], K, FP, out=dXin[:, p, :], in_drop=ctx.in_drop,
                                      seed=seed, row_offset=row_offset, seed_dev=seed_dev)
                if dres_in:
                    dXin[:, p, :] += dres_in[p]

        order = _path_order(streams, graphs, cfg.get("path_order"))
        for p in order:
            with _on_path(streams, p):
                rows_path(p)
        for p in order:
            with _on_path(streams, p):
                cols_path(p)
        _join(streams)
        if ctx.overlap is not None:
            ctx.overlap.join()
            ctx.overlap = None
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        ctx.saved_per_p = None
        if direct is not None:
            return (dXin,) + (None,) * 11
        return dXin, dW, da1, db1, da2, db2, dc, dWr, dbr, None, None, None


class WideHeadAttention(torch.autograd.Function):
    """K1 + K2 for ONE head wider than the 64 columns of a K1 / K2 row (hid_units > 64; models/gat.py:42-57 leaves
    the width free), every meta-path: the head runs as S = ceil(F'/64) column slices, each a K = 1, F' = 64 launch.

    What the slices of a head share is what makes them one head: the scores f1 / f2 (each slice's K1 epilogue gives
    its partial dot product, b1 / b2 ride in slice 0, the partials are added and K2 GATHERS f2 -- han_node_attn_fwd
    f2_src -- instead of recomputing it from its 64 columns), hence the coefficients, and the dropout draws keyed by
    the head: the per-head input dropout and the attention dropout (same seed, head index 0 in every slice).  The
    projected-row dropout is per column: slice s draws from its own stream (HAN_FLAG_FTS_SLICE).  The backward is the
    backward of the slices -- the softmax backward is linear in d alpha, so the slices' df1 / df2 add up -- with the
    totals in the places that need them (df1 into the transposed-graph pass, df2 into dH and the score gradients).

    forward(Xin, W (P,F,S*64), a1 (P,S*64), b1 (P,), a2 (P,S*64), b2 (P,), c (P,S*64), Wr, br, xs, graphs, cfg) -> M (N,P,S*64);
    columns beyond F' carry zero weights (their outputs are exactly act(0 + 0) and are cut off by the caller).
    cfg as for NodeLevelAttention.  Under a node partition every slice's table travels like a narrow head's (halo plan or
    all-gather, its own persistent exchange table) and so do the head's f2 totals (4 bytes per row); the forward is
    never replicated (xs_full is not used) and the backward exchanges one [g | stats] table per slice."""

    @staticmethod
    def forward(ctx, Xin, W, a1, b1, a2, b2, c, Wr, br, xs, graphs, cfg):
        P = len(graphs)
        if Xin is not None:
            Xin = Xin.contiguous()
            xs = tuple(Xin[:, p, :] for p in range(P))
        part = cfg.get("part")
        multi = part is not None and part.active
        row_offset = part.row_start if part is not None else 0
        plans_f = cfg.get("plans_f") if multi else None
        lay, grp = cfg.get("layer", 0), cfg.get("group", 0)
        S = W.shape[2] // D
        train = bool(cfg["train"])
        in_drop = float(cfg.get("in_drop", 0.0)) if train else 0.0
        coef_drop = float(cfg.get("coef_drop", 0.0)) if train else 0.0
        N, dev = xs[0].shape[0], W.device
        M = torch.empty((N, P, S * D), dtype=torch.float32, device=dev)
        seed_dev = cfg.get("seed_dev")
        tdt = cfg.get("table_dtype", torch.float32)
        zero1 = torch.zeros(1, dtype=torch.float32, device=dev)
        saved = []
        for p in range(P):
            seed = int(cfg["seeds"][p])
            Ws = W[p].view(-1, S, D).permute(1, 0, 2).contiguous()          # (S,F,64)
            a1s, a2s, cs = a1[p].view(S, 1, D), a2[p].view(S, 1, D), c[p].view(S, D)
            plan = plans_f[p] if plans_f is not None else None
            exchange = (plan.exchange_async if plan is not None else part.all_gather_rows_async) if multi else None
            Hs, Htabs, f1, f2, Rs = [], [], None, None, []
            for s_ in range(S):
                fl = ops.flag_fts_slice(s_)
                H, f1s, f2s = ops.project_fwd(xs[p], Ws[s_], a1s[s_], a2s[s_], b1[p:p + 1] if s_ == 0 else zero1,
                                              b2[p:p + 1] if s_ == 0 else zero1, in_drop=in_drop, fts_drop=in_drop,
                                              seed=seed, row_offset=row_offset, table_dtype=tdt, seed_dev=seed_dev, flags=fl)
                Hs.append(H)
                Htabs.append(exchange(H, ("wf", lay, grp, p, s_)) if multi else _Ready(H))
                f1 = f1s if f1 is None else f1 + f1s
                f2 = f2s if f2 is None else f2 + f2s
                R = None
                if Wr is not None:      # residual conv1d(seq, F', 1) of the DROPPED input (layers.py:38-40): same draws
                    Wrs = Wr[p].view(-1, S, D)[:, s_, :].contiguous()
                    R, _, _ = ops.project_fwd(xs[p], Wrs, a1s[s_], a2s[s_], zero1, zero1, in_drop=in_drop, fts_drop=0.0,
                                              seed=seed, row_offset=row_offset, seed_dev=seed_dev)
                    R = R + br[p].view(S, D)[s_]
                Rs.append(R)
            if cfg.get("coef_sink") is not None:
                if multi:
                    raise NotImplementedError("return_coef is not provided under a node partition")
                cfg["coef_sink"].append(ops.node_attn_coefs(graphs[p], f1, f2, coef_drop=coef_drop, seed=seed,
                                                            mean_heads=bool(cfg.get("coef_mean", False)),
                                                            seed_dev=seed_dev))
            f2_tab = exchange(f2, ("wf2", lay, grp, p)).wait() if multi else f2      # the head's scores of every table row
            per_s = []
            for s_ in range(S):
                _, sv = ops.node_attn_fwd(plan.graph if plan is not None else graphs[p], Htabs[s_].wait(), f1, a2s[s_],
                                          b2[p:p + 1], cs[s_],
                                          out=M[:, p, s_ * D:(s_ + 1) * D], train=train, coef_drop=coef_drop,
                                          fts_drop=in_drop, seed=seed, row_offset=row_offset, activation=cfg["act"],
                                          table_gid=plan.gid if plan is not None else None, res=Rs[s_],
                                          seed_dev=seed_dev, f2_src=f2_tab)
                if train:
                    per_s.append((Hs[s_], None) + sv[1:] + (Rs[s_],))      # the output slice: ctx.save_for_backward(M)
            if train:
                saved.append((f1, f2, per_s))
        ctx.cfg, ctx.xs, ctx.graphs, ctx.S = cfg, xs, graphs, S
        ctx.xin_shape = tuple(Xin.shape) if Xin is not None else None
        ctx.saved_per_p = saved
        ctx.in_drop, ctx.coef_drop = in_drop, coef_drop
        ctx.has_res = Wr is not None
        ctx.save_for_backward(W, a1, a2, c, *((Wr,) if Wr is not None else ()), *((M,) if train else ()))
        return M

    @staticmethod
    def backward(ctx, dM):
        W, a1, a2, c = ctx.saved_tensors[:4]
        Wr = ctx.saved_tensors[4] if ctx.has_res else None
        cfg, xs, graphs, S = ctx.cfg, ctx.xs, ctx.graphs, ctx.S
        if not cfg["train"]:
            raise RuntimeError("WideHeadAttention was run with train=False; no backward state")
        Mout = ctx.saved_tensors[-1]
        P, dev = len(graphs), W.device
        Fw = W.shape[1]
        dM = dM.contiguous()
        graphs_t = cfg.get("graphs_t") or tuple(g.transpose() for g in graphs)
        seed_dev = cfg.get("seed_dev")
        part = cfg.get("part")
        multi = part is not None and part.active
        row_offset = part.row_start if part is not None else 0
        plans_b = cfg.get("plans_b") if multi else None
        lay, grp = cfg.get("layer", 0), cfg.get("group", 0)
        dW = torch.empty((P, S, Fw, D), dtype=torch.float32, device=dev)
        da1, da2, dc = torch.empty_like(a1), torch.empty_like(a2), torch.empty_like(c)
        db1 = torch.empty((P,), dtype=torch.float32, device=dev)
        db2 = torch.empty((P,), dtype=torch.float32, device=dev)
        dWr = torch.empty((P, S, Fw, D), dtype=torch.float32, device=dev) if Wr is not None else None
        dXin = None
        if ctx.xin_shape is not None and ctx.needs_input_grad[0]:
            dXin = torch.zeros(ctx.xin_shape, dtype=torch.float32, device=dev)
        for p in range(P):
            f1, f2, per_s = ctx.saved_per_p[p]
            seed = int(cfg["seeds"][p])
            a1s, a2s, cs = a1[p].view(S, 1, D), a2[p].view(S, 1, D), c[p].view(S, D)
            plan = plans_b[p] if plans_b is not None else None
            exchange = (plan.exchange_async if plan is not None else part.all_gather_rows_async) if multi else None
            rows, df1 = [], None
            for s_ in range(S):       # row-local halves: g, the slice's share of df1
                H, _, lse, aggp, tsum, R = per_s[s_]
                gs, df1s, _ = ops.node_attn_bwd_rows(dM[:, p, s_ * D:(s_ + 1) * D], Mout[:, p, s_ * D:(s_ + 1) * D], aggp,
                                                     tsum, f1, lse, cs[s_],
                                                     activation=cfg["act"], K=1, FP=D, table_dtype=H.dtype, res=R,
                                                     dc_out=dc[p, s_ * D:(s_ + 1) * D])
                rows.append(exchange(gs, ("wb", lay, grp, p, s_)) if multi else _Ready(gs))      # one [g | stats] table per slice
                df1 = df1s if df1 is None else df1 + df1s
                if Wr is not None:
                    g32 = ops.gs_views(gs, 1, D, H.dtype)[0].to(torch.float32).contiguous()
                    ops.project_bwd(xs[p], g32, 1, D, in_drop=ctx.in_drop, seed=seed, row_offset=row_offset,
                                    seed_dev=seed_dev, out=dWr[p, s_])
                    if dXin is not None:
                        Wrs = Wr[p].view(-1, S, D)[:, s_, :].contiguous()
                        dXin[:, p, :] += ops.project_bwd_input(g32, Wrs, 1, D, in_drop=ctx.in_drop, seed=seed,
                                                               row_offset=row_offset, seed_dev=seed_dev)
            cols, df2 = [], None
            for s_ in range(S):       # transposed-graph halves with the head's df1; each returns its share of df2
                H = per_s[s_][0]
                dH, df2s = ops.node_attn_bwd_cols(plan.graph if plan is not None else graphs_t[p], rows[s_].wait(), H, f2,
                                                  df1, a1s[s_], a2s[s_],
                                                  coef_drop=ctx.coef_drop, fts_drop=ctx.in_drop, seed=seed,
                                                  src_offset=row_offset, dst_offset=0,
                                                  table_gid=plan.gid if plan is not None else None, seed_dev=seed_dev)
                cols.append((dH, df2s))
                df2 = df2s if df2 is None else df2 + df2s
            for s_ in range(S):
                H = per_s[s_][0]
                dH, df2s = cols[s_]
                if S > 1:             # the kernel added its own df2 share times a2; the head's total belongs there
                    dH.addcmul_(df2 - df2s, a2s[s_])
                o1 = torch.empty((1,), dtype=torch.float32, device=dev) if s_ else db1[p:p + 1]
                o2 = torch.empty((1,), dtype=torch.float32, device=dev) if s_ else db2[p:p + 1]
                ops.score_param_bwd(H, df1, df2, K=1, FP=D, out=(da1[p].view(S, 1, D)[s_], da2[p].view(S, 1, D)[s_], o1, o2))
                ops.project_bwd(xs[p], dH, 1, D, in_drop=ctx.in_drop, seed=seed, row_offset=row_offset, seed_dev=seed_dev,
                                out=dW[p, s_])
                if dXin is not None:
                    Ws = W[p].view(-1, S, D)[:, s_, :].contiguous()
                    dXin[:, p, :] += ops.project_bwd_input(dH, Ws, 1, D, in_drop=ctx.in_drop, seed=seed,
                                                           row_offset=row_offset, seed_dev=seed_dev)
        ctx.saved_per_p = None
        fold = lambda t: t.permute(0, 2, 1, 3).reshape(P, Fw, S * D)
        return (dXin, fold(dW), da1, db1, da2, db2, dc, fold(dWr) if dWr is not None else None,
                dc.clone() if Wr is not None else None, None, None, None)


class SemanticAttention(torch.autograd.Function):
    """K3: M (N,P,D) -> (Z (N,D), beta (N,P)); utils/layers.py:152-159."""

    @staticmethod
    def forward(ctx, M, w_omega, b_omega, u_omega):
        M = M.contiguous()
        Z, beta = ops.sem_attn_fwd(M, w_omega, b_omega, u_omega)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for beta (a fill launch per step)
        plist = (w_omega, b_omega, u_omega)
        ctx.direct = tuple(p.grad for p in plist) if all(_direct(p) for p in plist) else None
        ctx.save_for_backward(M, w_omega, b_omega, u_omega, beta)
        ctx.mark_non_differentiable(beta)
        return Z, beta

    @staticmethod
    def backward(ctx, dZ, _dbeta):
        M, w, b, u, beta = ctx.saved_tensors
        if dZ is None:
            dZ = torch.zeros((M.shape[0], M.shape[2]), dtype=M.dtype, device=M.device)
        dM, dw, db, du = ops.sem_attn_bwd(M, w, b, u, beta, dZ.contiguous(), out=ctx.direct)
        if ctx.direct is not None:
            return dM, None, None, None
        return dM, dw, db, du


class ClassifierLoss(torch.autograd.Function):
    """Fused classifier + masked softmax-CE (+ accuracy).
    models/gat.py:65-72, models/base_gattn.py:41-48,61-69.
    Returns (loss, accuracy, logits (N,C)); only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, Z, Wc, bc, labels, mask, row_weight, *grad_mode):
        # an eval forward under torch.no_grad() must neither pay for the gradient half of the kernel nor, in
        # direct-gradient mode, WRITE Wc.grad / bc.grad.  needs_input_grad reports requires_grad whatever the grad
        # mode and the grad mode is off inside every forward(), so the caller passes the one it was called under
        # (classifier_loss_any); without it the inputs decide, as before
        need = any(ctx.needs_input_grad[:3]) and (not grad_mode or bool(grad_mode[0]))
        ctx.n_opt = len(grad_mode)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the accuracy / logits outputs
        # direct-gradient mode additionally assumes the loss is the root of backward()
        # (d loss = 1), which is how HANTrainer calls it
        ctx.direct = need and _direct(Wc) and _direct(bc)
        logits, loss_acc, grads = ops.classifier_loss(Z.contiguous(), Wc, bc, labels, mask,
                                                      row_weight, backward=need,
                                                      grad_out=(Wc.grad, bc.grad) if ctx.direct else None)
        ctx.grads = grads
        if ctx.direct or not need:      # views of the kernel's output pair: no clone launches (eval forward: nothing to differentiate)
            loss, acc = loss_acc[0:1].view(()), loss_acc[1:2].view(())
        else:
            loss, acc = loss_acc[0].clone(), loss_acc[1].clone()
        ctx.mark_non_differentiable(acc, logits)
        return loss, acc, logits

    @staticmethod
    def backward(ctx, dloss, _dacc, _dlogits):
        dZ, dWc, dbc = ctx.grads
        ctx.grads = None
        tail = (None,) * (3 + ctx.n_opt)
        if ctx.direct:
            return (dZ, None, None) + tail
        if dloss is None:
            return (None, None, None) + tail
        return (dZ * dloss, dWc * dloss, dbc * dloss) + tail


class _ClassifierForward(torch.autograd.Function):
    """logits = (1/HC) sum_h (Z Wc[h] + bc[h]) (models/gat.py:65-72) by the HIP kernels, forward and backward
    (han_classifier_loss with an empty mask; han_classifier_bwd for a caller-supplied dlogits)."""

    @staticmethod
    def forward(ctx, Z, Wc, bc):
        N = Z.shape[0]
        labels = torch.zeros(N, dtype=torch.int32, device=Z.device)
        mask = torch.zeros(N, dtype=torch.uint8, device=Z.device)
        Z = Z.contiguous()
        logits, _, _ = ops.classifier_loss(Z, Wc, bc, labels, mask, 0.0, backward=False)
        ctx.save_for_backward(Z, Wc, bc)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        Z, Wc, bc = ctx.saved_tensors
        return ops.classifier_bwd(Z, Wc, bc, dlogits.contiguous())


def classifier(Z, Wc, bc):
    return _ClassifierForward.apply(Z, Wc, bc)


K3_WIDTHS = (64, 128)      # embedding widths the TUNED K3 / classifier kernels are built for
K3_MAX_ATT = 256           # attention sizes of the tuned K3 kernels: multiples of 64 up to 256
MAX_CLASSES = 64           # classes of the fused single-launch classifier kernels (more: the three-kernel path)


def _pad_to(n: int) -> int:
    return 64 * ((n + 63) // 64)


def semantic_attention(M, w_omega, b_omega, u_omega):
    """utils/layers.py:152-159 for any embedding width D and attention size A, always on the K3 kernels.
    Widths that are not multiples of 64 are zero-padded -- padded columns of w_omega / u_omega contribute
    tanh(.) * 0 = 0 to the scores and padded embedding columns are 0 -- which is exact.  D in {64, 128} with
    A <= 256 run the tuned kernels; anything wider (a last layer of 8 heads x 32, hid_units = [128],
    mp_att_size = 512, ...) the run-time-width kernels of sem_attn.hip (round 3: no torch branch is left)."""
    d, a = M.shape[2], w_omega.shape[1]
    dm, am = _pad_to(d), _pad_to(a)
    if dm == d and am == a:
        return SemanticAttention.apply(M.contiguous(), w_omega, b_omega, u_omega)
    Z, att = SemanticAttention.apply(
        torch.nn.functional.pad(M, (0, dm - d)), torch.nn.functional.pad(w_omega, (0, am - a, 0, dm - d)),
        torch.nn.functional.pad(b_omega, (0, am - a)), torch.nn.functional.pad(u_omega, (0, am - a)))
    return Z[:, :d], att


def classifier_any(Z, Wc, bc):
    """models/gat.py:65-72 for any embedding width and class count: always the HIP kernels (widths that are not
    multiples of 64 are zero-padded, which is exact)."""
    d = Z.shape[1]
    dm = _pad_to(d)
    if dm != d:
        Z, Wc = torch.nn.functional.pad(Z, (0, dm - d)), torch.nn.functional.pad(Wc, (0, 0, 0, dm - d))
    return classifier(Z, Wc, bc)


def classifier_loss_any(Z, Wc, bc, labels, mask, weight):
    """Classifier + masked softmax cross-entropy + accuracy (models/base_gattn.py:41-48,61-69) for any
    embedding width / class count; returns (loss, accuracy, logits)."""
    d = Z.shape[1]
    dm = _pad_to(d)
    if dm != d:
        Z, Wc = torch.nn.functional.pad(Z, (0, dm - d)), torch.nn.functional.pad(Wc, (0, 0, 0, dm - d))
    return ClassifierLoss.apply(Z, Wc, bc, labels, mask, weight, torch.is_grad_enabled())


# ---------------------------------------------------------------------------
# reference-named functional API
# ---------------------------------------------------------------------------
def _squeeze_batch(seq: torch.Tensor, name="seq") -> torch.Tensor:
    if seq.dim() == 3:
        if seq.shape[0] != 1:
            raise ValueError(f"{name}: batch size must be 1 (ex_acm3025.py:21; "
                             "utils/layers.py:110-113)")
        return seq[0]
    if seq.dim() != 2:
        raise ValueError(f"{name}: expected (1,N,F) or (N,F), got {tuple(seq.shape)}")
    return seq


def _single_head(seq, out_sz, graph, activation, in_drop, coef_drop, residual, params, training,
                 seed, return_coef=False):
    """One head of width out_sz through the D=64 kernels: the head occupies slot 0 of K = 64/F'k head
    slots of the lane-mapped width F'k = next of 4, 8, 16, 32, 64 >= out_sz; the columns beyond out_sz
    and the other slots have zero weights."""
    x = _squeeze_batch(seq)
    if out_sz < 1:
        raise ValueError("out_sz must be positive")
    if out_sz > D:
        return _single_wide_head(x, out_sz, graph, activation, in_drop, coef_drop, residual, params, training, seed,
                                 return_coef)
    fpk = next(w for w in (4, 8, 16, 32, 64) if out_sz <= w)
    K = D // fpk
    dev = x.device
    Fin = x.shape[1]

    def pad_last(t, width):
        out = t.new_zeros(t.shape[:-1] + (width,))
        out[..., :t.shape[-1]] = t
        return out

    W = pad_last(params["W"], D)[None]                                   # (1,F,D)
    a1 = torch.cat([pad_last(params["a1"], fpk)[None], x.new_zeros(K - 1, fpk)])[None]   # (1,K,F'k)
    a2 = torch.cat([pad_last(params["a2"], fpk)[None], x.new_zeros(K - 1, fpk)])[None]
    b1 = pad_last(params["b1"].reshape(1), K)[None]
    b2 = pad_last(params["b2"].reshape(1), K)[None]
    c = pad_last(params["c"], D)[None]
    if W.shape[1] != Fin:
        raise ValueError(f"W has {W.shape[1]} input features, seq has {Fin}")
    code, post = _act_code(activation)
    train = bool(training) or in_drop > 0 or coef_drop > 0 or W.requires_grad
    cfg = {"train": train, "in_drop": in_drop, "coef_drop": coef_drop,
           "seeds": (rng.next_seed() if seed is None else seed,), "act": code, "part": None,
           "coef_sink": [] if return_coef else None}
    Wr = br = None
    if residual and Fin != out_sz:
        # utils/layers.py:38-40: + conv1d(seq, out_sz, 1) of the DROPPED input, before the
        # activation; when the widths are equal the reference's branch is a no-op (:42)
        Wr = pad_last(params["res_W"], D)[None]
        br = pad_last(params["res_b"], D)[None]
    M = NodeLevelAttention.apply(None, W, a1, b1, a2, b2, c, Wr, br, (x,), (graph,), cfg)
    ret = M[:, 0, :out_sz]
    if post is not None:
        ret = post(ret)
    if return_coef:       # slot 0 of the K head slots is this head
        coefs = torch.sparse_csr_tensor(graph.rowptr, graph.colidx.long(),
                                        cfg["coef_sink"][0][:, 0].contiguous(), (graph.n_rows, graph.n_cols))
        return ret[None], coefs
    return ret[None]     # (1,N,out_sz)


def _single_wide_head(x, out_sz, graph, activation, in_drop, coef_drop, residual, params, training, seed,
                      return_coef):
    """One head wider than 64 columns: ceil(out_sz / 64) column slices that share the head's scores, coefficients
    and per-head dropout draws (WideHeadAttention); columns beyond out_sz carry zero weights."""
    S = -(-out_sz // D)
    Fin = x.shape[1]

    def pad_last(t):
        out = t.new_zeros(t.shape[:-1] + (S * D,))
        out[..., :t.shape[-1]] = t
        return out

    W = pad_last(params["W"])[None]
    if W.shape[1] != Fin:
        raise ValueError(f"W has {W.shape[1]} input features, seq has {Fin}")
    a1, a2, c = (pad_last(params[k])[None] for k in ("a1", "a2", "c"))
    b1, b2 = params["b1"].reshape(1), params["b2"].reshape(1)
    code, post = _act_code(activation)
    train = bool(training) or in_drop > 0 or coef_drop > 0 or W.requires_grad
    cfg = {"train": train, "in_drop": in_drop, "coef_drop": coef_drop,
           "seeds": (rng.next_seed() if seed is None else seed,), "act": code, "part": None,
           "coef_sink": [] if return_coef else None, "coef_mean": True}
    Wr = br = None
    if residual and Fin != out_sz:
        Wr, br = pad_last(params["res_W"])[None], pad_last(params["res_b"])[None]
    M = WideHeadAttention.apply(None, W, a1, b1, a2, b2, c, Wr, br, (x,), (graph,), cfg)
    ret = M[:, 0, :out_sz]
    if post is not None:
        ret = post(ret)
    if return_coef:
        coefs = torch.sparse_csr_tensor(graph.rowptr, graph.colidx.long(), cfg["coef_sink"][0].contiguous(),
                                        (graph.n_rows, graph.n_cols))
        return ret[None], coefs
    return ret[None]


def attn_head(seq, out_sz, bias_mat, activation, in_drop=0.0, coef_drop=0.0, residual=False,
              return_coef=False, *, params, training=False, seed=None):
    """utils/layers.py:7-46.  seq (1,N,F); bias_mat (1,N,N) additive mask, or a
    CSRGraph / (rowptr, colidx) pair.  params: dict W (F,out_sz), a1 (out_sz,),
    b1 (), a2 (out_sz,), b2 (), c (out_sz,) [+ res_W (F,out_sz), res_b (out_sz,) when
    residual=True and F != out_sz].
    return_coef=True also returns `coefs` (:43-44) -- the (dropped) softmax weights --
    as a torch sparse CSR tensor (N,N) over the stored neighbours (every masked entry
    of the reference's dense (1,N,N) tensor is exactly 0: exp(-1e9) == 0 in fp32);
    data only, no gradient flows through it."""
    return _single_head(seq, out_sz, as_graph(bias_mat, seq.device), activation, in_drop,
                        coef_drop, residual, params, training, seed, return_coef=return_coef)


def attn_head_const_1(seq, out_sz, bias_mat, activation, in_drop=0.0, coef_drop=0.0, residual=False, *,
                      params, training=False, seed=None):
    """utils/layers.py:49-81 (the HAN_nd ablation): logits := adjacency, i.e. every
    stored neighbour gets the same weight 1/deg (mean aggregator).  The same kernel
    with zero score parameters.  params: W (F,out_sz), c (out_sz,)."""
    z = torch.zeros_like(params["c"])
    p = {"W": params["W"], "a1": z, "a2": z, "b1": z.new_zeros(()), "b2": z.new_zeros(()), "c": params["c"]}
    for k in ("res_W", "res_b"):
        if k in params:
            p[k] = params[k]
    return attn_head(seq, out_sz, bias_mat, activation, in_drop=in_drop, coef_drop=coef_drop,
                     residual=residual, params=p, training=training, seed=seed)


def sp_attn_head(seq, out_sz, adj_mat, activation, nb_nodes, in_drop=0.0, coef_drop=0.0,
                 residual=False, *, params, training=False, seed=None):
    """utils/layers.py:85-127.  adj_mat: torch sparse (1,N,N)/(N,N) tensor, a
    CSRGraph or (rowptr, colidx).  Stored values other than 1 scale the logits,
    e_ij = LeakyReLU(v_ij*f1_i + v_ij*f2_j) (:95-98), as in the reference -- they
    are not a mask; the softmax runs over the stored entries (:100)."""
    g = as_graph(adj_mat, seq.device)
    if g.n_rows != nb_nodes:
        raise ValueError(f"nb_nodes={nb_nodes} but adj_mat has {g.n_rows} rows")
    return _single_head(seq, out_sz, g, activation, in_drop, coef_drop, residual, params, training,
                        seed)


def SimpleAttLayer(inputs, attention_size, time_major=False, return_alphas=False, *, params):
    """utils/layers.py:132-164.  inputs (N,P,D) [or (P,N,D) if time_major, or a
    tuple to be concatenated on the last axis]; params: w_omega (D,A),
    b_omega (A,), u_omega (A,)."""
    if isinstance(inputs, tuple):
        inputs = torch.cat(inputs, 2)                    # :134-136
    if time_major:
        inputs = inputs.transpose(0, 1)                  # :138-140
    if params["w_omega"].shape != (inputs.shape[2], attention_size):
        raise ValueError("w_omega must be (hidden_size, attention_size)")
    out, alphas = semantic_attention(inputs.contiguous(), params["w_omega"], params["b_omega"],
                                     params["u_omega"])
    if not return_alphas:
        return out
    return out, alphas
