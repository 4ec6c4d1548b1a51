"""Torch-CPU autograd restatement of the HAN hot path -- TEST INFRASTRUCTURE.

Second, independent restatement of the reference arithmetic (the first is
``oracle/han_oracle.py``) whose purpose is gradients (autograd, float64) and
the timed ``cpu_baseline`` leg of ``bench.py`` (float32, all host cores).
PARITY UNPINNED, for the reasons given in ``oracle/han_oracle.py``.

Parameters use the batched layout of the HIP implementation:
  W (P,F,D)  a1,a2 (P,K,F')  b1,b2 (P,K)  c (P,D)      D = K*F', column d = k*F'+f'
  w_omega (D,A)  b_omega (A,)  u_omega (A,)  Wc (Hc,D,C)  bc (Hc,C)
``to_batched`` converts from the per-head dicts of ``han_oracle.init_params``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline may
import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as Fnn

LEAKY_ALPHA = 0.2


def to_batched(params, dtype=torch.float64):
    """per-head dict params (han_oracle.init_params) -> batched torch tensors."""
    heads = params['heads']
    P, K = len(heads), len(heads[0])
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    out = {
        'W': torch.stack([torch.cat([t(h['W']) for h in hp], dim=1) for hp in heads]),
        'a1': torch.stack([torch.stack([t(h['a1']) for h in hp]) for hp in heads]),
        'a2': torch.stack([torch.stack([t(h['a2']) for h in hp]) for hp in heads]),
        'b1': torch.stack([torch.stack([t(h['b1']) for h in hp]) for hp in heads]),
        'b2': torch.stack([torch.stack([t(h['b2']) for h in hp]) for hp in heads]),
        'c': torch.stack([torch.cat([t(h['c']) for h in hp]) for hp in heads]),
        'w_omega': t(params['w_omega']), 'b_omega': t(params['b_omega']),
        'u_omega': t(params['u_omega']),
        'Wc': torch.stack([t(c['W']) for c in params['cls']]),
        'bc': torch.stack([t(c['b']) for c in params['cls']]),
    }
    assert out['W'].shape[0] == P and out['a1'].shape[1] == K
    if 'layers' in params:                      # extra node-attention layers, models/gat.py:48-57
        n_extra = len(params['layers'][0])
        for i in range(n_extra):
            lh = [params['layers'][p][i] for p in range(P)]
            sfx = f'_{i + 1}'
            out['W' + sfx] = torch.stack([torch.cat([t(h['W']) for h in hp], dim=1) for hp in lh])
            for nm in ('a1', 'a2', 'b1', 'b2'):
                out[nm + sfx] = torch.stack([torch.stack([t(h[nm]) for h in hp]) for hp in lh])
            out['c' + sfx] = torch.stack([torch.cat([t(h['c']) for h in hp]) for hp in lh])
            if 'res' in lh[0][0]:          # residual conv1d(seq, F', 1) per head (layers.py:38-40)
                out['Wr' + sfx] = torch.stack([torch.cat([t(h['res']['W']) for h in hp], dim=1) for hp in lh])
                out['br' + sfx] = torch.stack([torch.cat([t(h['res']['b']) for h in hp]) for hp in lh])
    return out


PARAM_ORDER = ('W', 'a1', 'b1', 'a2', 'b2', 'c', 'w_omega', 'b_omega', 'u_omega', 'Wc', 'bc')


def n_extra_layers(bp):
    n = 0
    while f'W_{n + 1}' in bp:
        n += 1
    return n


def param_order(bp):
    """PARAM_ORDER plus the extra layers' variables (after layer 0's, per layer)."""
    names = list(PARAM_ORDER[:6])
    for i in range(1, n_extra_layers(bp) + 1):
        names += [f'{nm}_{i}' for nm in ('W', 'a1', 'b1', 'a2', 'b2', 'c')]
        if f'Wr_{i}' in bp:
            names += [f'Wr_{i}', f'br_{i}']
    return tuple(names) + PARAM_ORDER[6:]


def node_attention_dense(x, bias_mat, W, a1, b1, a2, b2, c, keep_in=1.0, keep_coef=1.0,
                         masks=None, Wr=None, br=None):
    """All K heads of one meta-path, dense additive-mask form.
    utils/layers.py:18-35,46 per head; head concat models/gat.py:46.
    x (N,F); bias_mat (N,N); W (F,D); a1,a2 (K,F'); b1,b2 (K,); c (D,).
    masks: None or dict 'seq' (K,N,F), 'coef' (K,N,N), 'fts' (N,D) of {0,1}.
    Returns (N,D) = ELU(coefs @ H + c).
    """
    K, Fp = a1.shape
    outs = []
    for k in range(K):
        xs = x
        if masks is not None and 'seq' in masks:
            xs = x / keep_in * masks['seq'][k]                         # layers.py:19
        h = xs @ W[:, k * Fp:(k + 1) * Fp]                             # :20
        f1 = h @ a1[k] + b1[k]                                         # :23
        f2 = h @ a2[k] + b2[k]                                         # :24
        logits = f1[:, None] + f2[None, :]                             # :26
        coefs = torch.softmax(Fnn.leaky_relu(logits, LEAKY_ALPHA) + bias_mat, dim=-1)  # :27
        if masks is not None and 'coef' in masks:
            coefs = coefs / keep_coef * masks['coef'][k]               # :30
        if masks is not None and 'fts' in masks:
            h = h / keep_in * masks['fts'][:, k * Fp:(k + 1) * Fp]     # :32
        vals = coefs @ h                                               # :34
        ret = vals + c[k * Fp:(k + 1) * Fp]                            # :35
        if Wr is not None:                                             # :38-40 (seq is the dropped input)
            ret = ret + xs @ Wr[:, k * Fp:(k + 1) * Fp] + br[k * Fp:(k + 1) * Fp]
        outs.append(Fnn.elu(ret))                                      # :46
    return torch.cat(outs, dim=-1)


class _StoreBF16(torch.autograd.Function):
    """The bf16 storage mode of the HIP path, restated (it has no counterpart in the reference, which is
    fp32 throughout): a projected row is rounded to bf16 (nearest-even) when K1 stores it and, in training,
    the LOWEST mantissa bit of each element is overwritten with its projected-row-dropout keep bit
    (han_amd/csrc/project.hip); every consumer reads that stored value.  Straight-through gradient."""

    @staticmethod
    def forward(ctx, h, keepbits):
        b = h.to(torch.float32).to(torch.bfloat16).view(torch.int16)
        if keepbits is not None:
            b = (b & -2) | keepbits.to(torch.int16)
        return b.view(torch.bfloat16).to(h.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundGradBF16(torch.autograd.Function):
    """Identity whose incoming gradient is rounded to bf16: the backward's g = dOut * act' table is
    stored in bf16 and both halves of the K2 backward read the stored value (node_attn.hip)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32).to(torch.bfloat16).to(g.dtype)


def node_attention_csr(x, rowptr, colidx, W, a1, b1, a2, b2, c, keep_in=1.0, keep_coef=1.0,
                       masks=None, adj_vals=None, Wr=None, br=None, table_bf16=False):
    """All K heads of one meta-path over CSR neighbours only -- what
    sp_attn_head (utils/layers.py:85-127) computes; heads batched.
    masks: 'seq' (K,N,F), 'coef' (E,K), 'fts' (N,D).
    table_bf16: emulate the bf16 storage of the H / g tables (see _StoreBF16, _RoundGradBF16).
    """
    N = x.shape[0]
    K, Fp = a1.shape
    D = K * Fp
    deg = rowptr[1:] - rowptr[:-1]
    rows = torch.repeat_interleave(torch.arange(N), deg)
    cols = colidx.long()
    if masks is not None and 'seq' in masks:
        h = torch.cat([(x / keep_in * masks['seq'][k]) @ W[:, k * Fp:(k + 1) * Fp]
                       for k in range(K)], dim=1)                       # :87-90 per head
    else:
        h = x @ W                                                       # :90
    if table_bf16:
        h = _StoreBF16.apply(h, masks['fts'] if (masks is not None and 'fts' in masks) else None)
    hk = h.view(N, K, Fp)
    f1 = (hk * a1[None]).sum(-1) + b1[None]                             # :93  (N,K)
    f2 = (hk * a2[None]).sum(-1) + b2[None]                             # :94
    if adj_vals is None:
        lg = f1[rows] + f2[cols]                                        # :95-96 binary adj
    else:
        lg = adj_vals[:, None] * f1[rows] + adj_vals[:, None] * f2[cols]
    lg = Fnn.leaky_relu(lg, LEAKY_ALPHA)                                # :97-99  (E,K)
    mx = torch.full((N, K), -float('inf'), dtype=lg.dtype)
    mx = mx.scatter_reduce(0, rows[:, None].expand(-1, K), lg, reduce='amax')
    ex = torch.exp(lg - mx[rows])
    den = torch.zeros((N, K), dtype=lg.dtype).index_add(0, rows, ex)
    coefs = ex / den[rows]                                              # :100
    if masks is not None and 'coef' in masks:
        coefs = coefs / keep_coef * masks['coef']                       # :102-106
    if masks is not None and 'fts' in masks:
        h = h / keep_in * masks['fts']                                  # :107-108
    msg = coefs[:, :, None] * h.view(N, K, Fp)[cols]                    # (E,K,F')
    vals = torch.zeros((N, K, Fp), dtype=h.dtype).index_add(0, rows, msg)   # :113
    if table_bf16:
        vals = _RoundGradBF16.apply(vals)
    ret = vals.reshape(N, D) + c                                        # :118
    if Wr is not None:                                                  # :121-123
        if masks is not None and 'seq' in masks:
            ret = ret + torch.cat([(x / keep_in * masks['seq'][k]) @ Wr[:, k * Fp:(k + 1) * Fp]
                                   for k in range(K)], dim=1) + br
        else:
            ret = ret + x @ Wr + br
    return Fnn.elu(ret)                                                 # :127


def semantic_attention(m, w_omega, b_omega, u_omega):
    """utils/layers.py:152-159.  m (N,P,D) -> (N,D), (N,P)."""
    v = torch.tanh(m @ w_omega + b_omega)
    vu = v @ u_omega
    alphas = torch.softmax(vu, dim=-1)
    return (m * alphas[..., None]).sum(1), alphas


def hetegat_forward(x_list, graphs, bp, keep_in=1.0, keep_coef=1.0, masks=None, dense=False, table_bf16=False):
    """models/gat.py:34-77 with hid_units=[F'], batched heads.
    x_list[p] (N,F); graphs[p] = bias_mat (N,N) if dense else (rowptr, colidx).
    Returns logits (N,C), final_embed (N,D), att_val (N,P)."""
    embeds = []
    for p, (x, g) in enumerate(zip(x_list, graphs)):                    # gat.py:39
        mk = masks[p] if masks is not None else None
        args = (bp['W'][p], bp['a1'][p], bp['b1'][p], bp['a2'][p], bp['b2'][p], bp['c'][p])
        if dense:
            e = node_attention_dense(x, g, *args, keep_in=keep_in, keep_coef=keep_coef, masks=mk)
        else:
            e = node_attention_csr(x, g[0], g[1], *args, keep_in=keep_in, keep_coef=keep_coef,
                                   masks=mk, table_bf16=table_bf16)
        for i in range(1, n_extra_layers(bp) + 1):                      # gat.py:48-57
            sfx = f'_{i}'
            largs = tuple(bp[nm + sfx][p] for nm in ('W', 'a1', 'b1', 'a2', 'b2', 'c'))
            lmk = mk['layers'][i - 1] if (mk is not None and 'layers' in mk) else None
            res = dict(Wr=bp['Wr' + sfx][p], br=bp['br' + sfx][p]) if ('Wr' + sfx) in bp else {}
            if dense:
                e = node_attention_dense(e, g, *largs, keep_in=keep_in, keep_coef=keep_coef, masks=lmk, **res)
            else:
                e = node_attention_csr(e, g[0], g[1], *largs, keep_in=keep_in, keep_coef=keep_coef,
                                       masks=lmk, **res)
        embeds.append(e[:, None, :])                                    # gat.py:58
    m = torch.cat(embeds, dim=1)                                        # gat.py:60
    final_embed, att = semantic_attention(m, bp['w_omega'], bp['b_omega'], bp['u_omega'])
    hc = bp['Wc'].shape[0]
    logits = sum(final_embed @ bp['Wc'][i] + bp['bc'][i] for i in range(hc)) / hc   # gat.py:66-72
    return logits, final_embed, att


def masked_softmax_cross_entropy(logits, labels, mask):
    """models/base_gattn.py:41-48."""
    loss = -(labels * torch.log_softmax(logits, dim=-1)).sum(-1)
    mask = mask.to(logits.dtype)
    mask = mask / mask.mean()
    return (loss * mask).mean()


def masked_accuracy(logits, labels, mask):
    """models/base_gattn.py:61-69."""
    correct = (logits.argmax(1) == labels.argmax(1)).to(logits.dtype)
    mask = mask.to(logits.dtype)
    mask = mask / mask.mean()
    return (correct * mask).mean()


def l2_all(bp, l2_coef):
    """models/base_gattn.py:14-16 (applies to every trainable, biases included)."""
    return l2_coef * sum((bp[k] ** 2).sum() / 2 for k in PARAM_ORDER)


def adam_step_tf_(bp, grads, state, lr=0.005, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place tf.train.AdamOptimizer step (models/base_gattn.py:19-22)."""
    state['t'] += 1
    t = state['t']
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for k in PARAM_ORDER:
            m, v = state['m'][k], state['v'][k]
            m.mul_(beta1).add_(grads[k], alpha=1 - beta1)
            v.mul_(beta2).addcmul_(grads[k], grads[k], value=1 - beta2)
            bp[k].sub_(lr_t * m / (v.sqrt() + eps))


def new_adam_state(bp):
    return {'t': 0, 'm': {k: torch.zeros_like(bp[k]) for k in PARAM_ORDER},
            'v': {k: torch.zeros_like(bp[k]) for k in PARAM_ORDER}}


def train_epoch(x_list, graphs, bp, state, labels, train_mask, val_mask, lr=0.005,
                l2_coef=0.001, keep=0.4, masks=None, dense=False):
    """One reference epoch (ex_acm3025.py:171-218): one fwd+bwd+Adam step with
    dropout 0.6/0.6 on the train mask, then one eval forward on the val mask."""
    for k in PARAM_ORDER:
        bp[k].requires_grad_(True)
        bp[k].grad = None
    logits, _, _ = hetegat_forward(x_list, graphs, bp, keep_in=keep, keep_coef=keep,
                                   masks=masks, dense=dense)
    loss = masked_softmax_cross_entropy(logits, labels, train_mask) + l2_all(bp, l2_coef)
    loss.backward()
    grads = {k: bp[k].grad for k in PARAM_ORDER}
    for k in PARAM_ORDER:
        bp[k].requires_grad_(False)
    adam_step_tf_(bp, grads, state, lr=lr)
    with torch.no_grad():
        vlogits, _, _ = hetegat_forward(x_list, graphs, bp, dense=dense)
        vloss = masked_softmax_cross_entropy(vlogits, labels, val_mask)
        vacc = masked_accuracy(vlogits, labels, val_mask)
    return float(loss.detach()), float(vloss), float(vacc)
