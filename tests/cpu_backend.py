"""CPU stand-ins for ``han_amd.ops`` -- TEST INFRASTRUCTURE ONLY.

``install()`` monkey-patches the functions of ``han_amd.ops`` with torch-CPU
restatements of each kernel's CONTRACT (same arguments, same outputs, same
dropout RNG streams), so that the host logic above the C ABI -- autograd wiring,
flat parameter/gradient buffers, HANTrainer, NodePartition and its collectives --
runs under ``gloo`` with world_size 2 on a machine without a GPU.  The product
never imports this module and has no CPU path of its own.

The formulas are the hand-derived backward of SURVEY.md section 8a; that they
agree with float64 autograd of the oracle is itself tested (test_host_cpu.py).
"""
import numpy as np
import torch

from tests import rng_ref

D = 64
SLOPE = 0.2


def _eff(seed, seed_dev):
    return rng_ref.resolve_seed(seed, None if seed_dev is None else int(seed_dev.item()) & ((1 << 64) - 1))


def _f64(t):
    return t.detach().to(torch.float64)


def _rows_of(graph):
    return torch.repeat_interleave(torch.arange(graph.n_rows), graph.degrees())


def project_fwd(X, W, a1, a2, b1, b2, in_drop=0.0, fts_drop=0.0, seed=0, row_offset=0,
                table_dtype=torch.float32, seed_dev=None, flags=0, want_keep=False):
    seed = _eff(seed, seed_dev)
    N, F = X.shape
    K, FP = a1.shape
    x, w = _f64(X), _f64(W)
    if in_drop > 0:
        keep = rng_ref.keep_prob32(in_drop)
        sm = torch.tensor(rng_ref.seq_mask(seed, N, F, K, in_drop, row_offset))
        H = torch.cat([(x / keep * sm[k]) @ w[:, k * FP:(k + 1) * FP] for k in range(K)], 1)
    else:
        H = x @ w
    H = H.to(torch.float32)
    it = torch.int32
    if table_dtype == torch.bfloat16:       # bf16 storage: round to nearest even, keep bit = bit 0 of the bf16
        H, it = H.to(torch.bfloat16), torch.int16
    if fts_drop > 0:      # keep bits ride in mantissa bit 0 (han_project_fwd contract)
        bits = torch.tensor(rng_ref.fts_mask(seed, N, D, fts_drop, row_offset, (int(flags) >> 8) & 0xFF)).to(it)
        H = ((H.view(it) & ~1) | bits).view(H.dtype)
    hk = _f64(H).view(N, K, FP)
    f1 = (hk * _f64(a1)[None]).sum(-1) + _f64(b1)
    f2 = (hk * _f64(a2)[None]).sum(-1) + _f64(b2)
    if want_keep:       # the keep table exists for large inputs only: the host logic must cope with None
        return H, f1.to(torch.float32), f2.to(torch.float32), None
    return H, f1.to(torch.float32), f2.to(torch.float32)


def project_fwd_multi(X, W, a1, a2, b1, b2, in_drop=0.0, fts_drop=0.0, seeds=None, row_offset=0,
                      table_dtype=torch.float32, seed_dev=None, flags=0, want_keep=False):
    P = W.shape[0]
    r = [project_fwd(X, W[p], a1[p], a2[p], b1[p], b2[p], in_drop, fts_drop, seeds[p] if seeds is not None else 0,
                     row_offset, table_dtype, seed_dev) for p in range(P)]
    out = tuple(torch.stack([v[i] for v in r]) for i in range(3))
    return out + ([None] * P,) if want_keep else out


def keep_bytes(N, F, ldx, K=8, FP=8):
    return 0


def _put(out, val):
    if out is None:
        return val
    out.copy_(val)
    return out


def project_bwd(X, dH, K, FP, in_drop=0.0, seed=0, row_offset=0, seed_dev=None, out=None, keep=None):
    seed = _eff(seed, seed_dev)
    N, F = X.shape
    x, d = _f64(X), _f64(dH)
    if in_drop > 0:
        keep = rng_ref.keep_prob32(in_drop)
        sm = torch.tensor(rng_ref.seq_mask(seed, N, F, K, in_drop, row_offset))
        dW = torch.cat([(x / keep * sm[k]).t() @ d[:, k * FP:(k + 1) * FP] for k in range(K)], 1)
    else:
        dW = x.t() @ d
    return _put(out, dW.to(torch.float32))


def project_bwd_input(dH, W, K, FP, out=None, in_drop=0.0, seed=0, row_offset=0, seed_dev=None):
    seed = _eff(seed, seed_dev)
    N, F = dH.shape[0], W.shape[0]
    d, w = _f64(dH), _f64(W)
    if in_drop > 0:
        keep = rng_ref.keep_prob32(in_drop)
        sm = torch.tensor(rng_ref.seq_mask(seed, N, F, K, in_drop, row_offset))
        dX = sum(sm[k] / keep * (d[:, k * FP:(k + 1) * FP] @ w[:, k * FP:(k + 1) * FP].t()) for k in range(K))
    else:
        dX = d @ w.t()
    if out is None:
        out = torch.empty((N, F), dtype=torch.float32)
    out.copy_(dX.to(torch.float32))
    return out


def _edge_terms(graph, H_tab, f1, a2, b2, coef_drop, fts_drop, seed, row_offset, table_gid=None, f2_src=None):
    K, FP = a2.shape
    rows = _rows_of(graph)
    cols = graph.colidx.long()
    h = _f64(H_tab)
    f2 = (h.view(-1, K, FP) * _f64(a2)[None]).sum(-1) + _f64(b2) if f2_src is None else _f64(f2_src)
    u = _f64(f1)[rows] + f2[cols]                                  # (E,K)
    w = _f64(graph.values)[:, None] if graph.values is not None else torch.ones((1, 1), dtype=torch.float64)
    u = w * u                                                      # layers.py:95-96
    sg = torch.where(u > 0, torch.ones_like(u), torch.full_like(u, SLOPE)) * w
    e = torch.maximum(u, SLOPE * u)
    mx = torch.full((graph.n_rows, K), -1e30, dtype=torch.float64)
    mx = mx.scatter_reduce(0, rows[:, None].expand(-1, K), e, reduce="amax")
    ex = torch.exp(e - mx[rows])
    den = torch.zeros((graph.n_rows, K), dtype=torch.float64).index_add(0, rows, ex)
    alpha = ex / den[rows]
    lse = mx + torch.log(den)
    am = torch.ones_like(alpha)
    if coef_drop > 0:
        keep = rng_ref.keep_prob32(coef_drop)
        cols_g = graph.colidx.numpy() if table_gid is None else table_gid.numpy()[graph.colidx.numpy()]
        am = torch.tensor(rng_ref.coef_mask_csr(seed, graph.rowptr.numpy(), cols_g, K,
                                                coef_drop, row_offset)) / keep
    hd = h
    if fts_drop > 0:
        keepf = rng_ref.keep_prob32(fts_drop)
        bits = (H_tab.view(torch.int16 if H_tab.dtype == torch.bfloat16 else torch.int32) & 1).to(torch.float64)
        hd = h * bits / keepf
    return rows, cols, alpha, am, sg, lse, hd


def node_attn_fwd(graph, H_tab, f1, a2, b2, c, out=None, train=False, coef_drop=0.0, fts_drop=0.0,
                  seed=0, row_offset=0, activation=1, table_gid=None, res=None, seed_dev=None, f2_src=None, f2=None):
    seed = _eff(seed, seed_dev)
    K, FP = a2.shape
    N = graph.n_rows
    rows, cols, alpha, am, sg, lse, hd = _edge_terms(graph, H_tab, f1, a2, b2, coef_drop, fts_drop,
                                                     seed, row_offset, table_gid, f2_src)
    hk = hd.view(-1, K, FP)[cols]                                    # (E,K,FP)
    agg = torch.zeros((N, K, FP), dtype=torch.float64).index_add(0, rows, (alpha * am)[:, :, None] * hk)
    pre = agg.reshape(N, D) + _f64(c)
    if res is not None:
        pre = pre + _f64(res)
    o = torch.where(pre > 0, pre, torch.expm1(pre)) if activation == 1 else pre
    if out is None:
        out = torch.empty((N, D), dtype=torch.float32)
    out.copy_(o.to(torch.float32))
    saved = None
    if train:
        aggp = torch.zeros((N, K, FP), dtype=torch.float64).index_add(
            0, rows, (alpha * am * sg)[:, :, None] * hk).reshape(N, D)
        tsum = torch.zeros((N, K), dtype=torch.float64).index_add(0, rows, alpha * sg)
        empty = graph.degrees() == 0
        lse = torch.where(empty[:, None], torch.full_like(lse, -1e30), lse)
        saved = (out,) + tuple(t.to(torch.float32) for t in (lse, aggp, tsum))      # saved[0] = the output view (ops contract)
    return out, saved


def _gs_row_bytes(K, table_dtype=torch.float32):
    gb = D * (2 if table_dtype == torch.bfloat16 else 4)
    return ((gb + 16 * K + 127) // 128) * 128


def gs_row_bytes(K=8, FP=8, table_dtype=torch.float32):
    return _gs_row_bytes(K, table_dtype)


def gs_views(gs, K=8, FP=8, table_dtype=torch.float32):
    gb = D * (2 if table_dtype == torch.bfloat16 else 4)
    return gs[:, :gb].view(table_dtype), gs[:, gb:gb + 16 * K].view(torch.float32).unflatten(1, (K, 4))


def node_attn_bwd_rows(dOut, out, aggp, tsum, f1, lse, c, activation=1, K=8, FP=8,
                       table_dtype=torch.float32, res=None, dc_out=None, gs_out=None):
    N = out.shape[0]
    o = _f64(out)
    if activation == 1:       # the pre-activation from the output: ELU inverted (han_node_attn_bwd_rows)
        da = torch.where(o <= 0, o + 1.0, torch.ones_like(o))
        p = torch.where(o <= 0, torch.log(da.clamp_min(1e-300)), o)
        p = torch.where(da > 0, p, torch.zeros_like(p))
    else:
        da, p = torch.ones_like(o), o
    g = _f64(dOut) * da
    dc = g.sum(0)
    g = _f64(g.to(torch.float32).to(table_dtype))     # the stored g is what both backward halves use
    agg = p - _f64(c) - (_f64(res) if res is not None else 0.0)
    s = (g * agg).view(N, K, FP).sum(-1)
    dp = (g * _f64(aggp)).view(N, K, FP).sum(-1)
    df1 = dp - s * _f64(tsum)
    stats = torch.stack([_f64(f1), _f64(lse), s, torch.zeros_like(s)], dim=-1)
    gs = gs_out if gs_out is not None else torch.zeros((N, _gs_row_bytes(K, table_dtype)), dtype=torch.uint8)
    gv, sv = gs_views(gs, K, FP, table_dtype)
    gv.copy_(g.to(table_dtype))
    sv.copy_(stats.to(torch.float32))
    return gs, df1.to(torch.float32), _put(dc_out, dc.to(torch.float32))


def node_attn_bwd_cols(graph_t, gs_tab, H, f2, df1, a1, a2, coef_drop=0.0, fts_drop=0.0,
                       seed=0, src_offset=0, dst_offset=0, table_gid=None, seed_dev=None):
    seed = _eff(seed, seed_dev)
    K, FP = a1.shape
    g_tab, stats_tab = gs_views(gs_tab, K, FP, H.dtype)
    NS = graph_t.n_rows
    src = _rows_of(graph_t)                       # local source j per transposed edge
    dst = graph_t.colidx.long()                   # destination i (table index)
    st = _f64(stats_tab)
    w = _f64(graph_t.values)[:, None] if graph_t.values is not None else torch.ones((1, 1), dtype=torch.float64)
    if getattr(graph_t, "masked", False):         # HAN_FLAG_MASKED_EDGES: negative entries are skipped
        live = dst >= 0
        src, dst = src[live], dst[live]
        if graph_t.values is not None:
            w = w[live]
    u = w * (st[dst, :, 0] + _f64(f2)[src])
    sg = torch.where(u > 0, torch.ones_like(u), torch.full_like(u, SLOPE)) * w
    alpha = torch.exp(torch.maximum(u, SLOPE * u) - st[dst, :, 1])
    am = torch.ones_like(alpha)
    if coef_drop > 0:
        keep = rng_ref.keep_prob32(coef_drop)
        dst_g = dst.numpy() if table_gid is None else table_gid.numpy()[dst.numpy()]
        am = torch.tensor(rng_ref.coef_draws(seed, dst_g + dst_offset, src.numpy() + src_offset,
                                             K, coef_drop)) / keep
    h64 = _f64(H)
    mk = torch.ones_like(h64)
    if fts_drop > 0:
        mk = (H.view(torch.int16 if H.dtype == torch.bfloat16 else torch.int32) & 1).to(torch.float64) \
            / rng_ref.keep_prob32(fts_drop)
    hd = (h64 * mk).view(NS, K, FP)
    gk = _f64(g_tab).reshape(-1, K, FP)[dst]                       # (E,K,FP)
    dot = (gk * hd[src]).sum(-1)
    dl = alpha * sg * (am * dot - st[dst, :, 2])
    df2 = torch.zeros((NS, K), dtype=torch.float64).index_add(0, src, dl)
    acc = torch.zeros((NS, K, FP), dtype=torch.float64).index_add(0, src, (alpha * am)[:, :, None] * gk)
    dH = acc * mk.view(NS, K, FP) + _f64(df1)[:, :, None] * _f64(a1)[None] + df2[:, :, None] * _f64(a2)[None]
    return dH.reshape(NS, D).to(torch.float32), df2.to(torch.float32)


def score_param_bwd(H, df1, df2, K=8, FP=8, out=None):
    hk = _f64(H).view(-1, K, FP)
    da1 = (_f64(df1)[:, :, None] * hk).sum(0)
    da2 = (_f64(df2)[:, :, None] * hk).sum(0)
    res = (da1.to(torch.float32), da2.to(torch.float32), _f64(df1).sum(0).to(torch.float32),
           _f64(df2).sum(0).to(torch.float32))
    o = out if out is not None else (None,) * 4
    return tuple(_put(o[i], res[i]) for i in range(4))


def sem_attn_fwd(M, w_omega, b_omega, u_omega, flags=0):
    m = _f64(M)
    v = torch.tanh(m @ _f64(w_omega) + _f64(b_omega))
    beta = torch.softmax(v @ _f64(u_omega), dim=-1)
    return (m * beta[..., None]).sum(1).to(torch.float32), beta.to(torch.float32)


def sem_attn_bwd(M, w_omega, b_omega, u_omega, beta, dZ, out=None, flags=0):
    m, w, b, u = (_f64(t).requires_grad_(True) for t in (M, w_omega, b_omega, u_omega))
    with torch.enable_grad():
        v = torch.tanh(m @ w + b)
        z = (m * torch.softmax(v @ u, dim=-1)[..., None]).sum(1)
        (z * _f64(dZ)).sum().backward()
    res = tuple(t.grad.to(torch.float32) for t in (m, w, b, u))
    o = out if out is not None else (None,) * 3
    return (res[0],) + tuple(_put(o[i], res[i + 1]) for i in range(3))


def classifier_loss(Z, Wc, bc, labels, mask, row_weight, backward=False, grad_out=None):
    z, w, b = (_f64(t).requires_grad_(True) for t in (Z, Wc, bc))
    hc = w.shape[0]
    with torch.enable_grad():
        logits = sum(z @ w[i] + b[i] for i in range(hc)) / hc
        wt = mask.to(torch.float64) * row_weight
        logp = torch.log_softmax(logits, dim=-1)
        ce = -logp.gather(1, labels.long()[:, None])[:, 0]
        loss = (wt * ce).sum()
        acc = (wt * (logits.argmax(1) == labels.long()).to(torch.float64)).sum()
        grads = None
        if backward:
            loss.backward()
            go = grad_out if grad_out is not None else (None, None)
            grads = (z.grad.to(torch.float32), _put(go[0], w.grad.to(torch.float32)),
                     _put(go[1], b.grad.to(torch.float32)))
    la = torch.stack([loss.detach(), acc]).to(torch.float32)
    return logits.detach().to(torch.float32), la, grads


def classifier_bwd(Z, Wc, bc, dlogits):
    z, w, dl = _f64(Z), _f64(Wc), _f64(dlogits)
    hc = w.shape[0]
    dZ = dl @ w.mean(0).t()
    dWc = (z.t() @ dl / hc)[None].expand(hc, -1, -1).contiguous()
    dbc = (dl.sum(0) / hc)[None].expand(hc, -1).contiguous()
    return dZ.to(torch.float32), dWc.to(torch.float32), dbc.to(torch.float32)


def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, l2_coef=0.0, step_dev=None):
    if step_dev is not None:
        t = int(step_dev.item())
        lr_t = float(np.float32(lr_t * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)))
    g = grad + l2_coef * param
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    param.sub_(lr_t * m / (v.sqrt() + eps))


def l2_half_sumsq(param):
    return (param.double() ** 2).sum().mul(0.5).to(torch.float32).reshape(1)


def require_gpu(t, name):
    return t


_NAMES = ("require_gpu", "project_fwd", "project_fwd_multi", "keep_bytes", "project_bwd", "project_bwd_input", "node_attn_fwd", "node_attn_bwd_rows", "node_attn_bwd_cols",
          "gs_row_bytes", "gs_views",
          "score_param_bwd", "sem_attn_fwd", "sem_attn_bwd", "classifier_loss", "classifier_bwd", "adam_step",
          "l2_half_sumsq")


def install():
    """Patch han_amd.ops in THIS process (tests only)."""
    from han_amd import ops
    for n in _NAMES:
        setattr(ops, n, globals()[n])
