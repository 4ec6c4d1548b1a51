"""CPU oracle for the HAN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``han_amd/`` imports it; the product
path fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference delegates all arithmetic to TensorFlow 1.x, which
is not installed here and cannot be installed (no network); the reference holds
no tests, golden vectors or fixtures for this path.  This
file is therefore a NumPy restatement, op for op, of the reference's *call
sites*, with TF op semantics taken from the TF1 documentation.  The one piece
of the reference that runs here (``utils/process.py:14-25`` ``adj_to_bias``,
NumPy only) pins :func:`adj_to_bias` through ``tests/golden/adj_to_bias_*.npz``.

Every function cites the reference lines it follows (paths relative to
/root/reference).  All arithmetic is done in the dtype of the inputs: feed
float64 for an oracle, float32 to mimic the reference's precision.
"""
from __future__ import annotations

import numpy as np

LEAKY_ALPHA = 0.2      # tf.nn.leaky_relu default, used at utils/layers.py:27,98
MASK_VALUE = -1e9      # utils/process.py:25


# --------------------------------------------------------------------------
# TF1 primitive semantics (third-party, restated from documentation)
# --------------------------------------------------------------------------
def leaky_relu(x, alpha=LEAKY_ALPHA):
    """tf.nn.leaky_relu: max(x, alpha*x) for alpha < 1."""
    return np.maximum(x, alpha * x)


def elu(x):
    """tf.nn.elu: x if x > 0 else exp(x) - 1."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def softmax(x, axis=-1):
    """tf.nn.softmax: max-subtracted exponentials, normalised over `axis`."""
    z = x - np.max(x, axis=axis, keepdims=True)
    e = np.exp(z)
    return e / np.sum(e, axis=axis, keepdims=True)


def dropout_apply(x, keep_prob, mask):
    """tf.nn.dropout(x, keep_prob) with the Bernoulli draw made explicit.

    TF computes ``x / keep_prob * floor(keep_prob + U[0,1))``; ``mask`` is that
    floor term (1 = kept).  ``mask=None`` means keep_prob == 1 (eval feed of
    0.0 drop, ex_acm3025.py:208-209).
    """
    if mask is None:
        return x
    return x / keep_prob * mask


def glorot_uniform(rng, fan_in, fan_out, shape, dtype=np.float64):
    """tf.layers default kernel initializer (glorot_uniform)."""
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)


# --------------------------------------------------------------------------
# utils/process.py:14-25
# --------------------------------------------------------------------------
def adj_to_bias(adj, sizes, nhood=1):
    """Restates utils/process.py:14-25.

    adj: (G, N, N).  mt = (adj + I)^nhood; entries > 0 inside [0:sizes[g]]^2 are
    set to 1 (entries outside that square are left as computed -- the reference
    does the same); returns -1e9 * (1 - mt) in float64.
    """
    adj = np.asarray(adj)
    nb_graphs, n = adj.shape[0], adj.shape[1]
    mt = np.empty(adj.shape)                                  # :16 (float64)
    for g in range(nb_graphs):
        mt[g] = np.eye(n)                                     # :18
        for _ in range(nhood):                                # :19-20
            mt[g] = np.matmul(mt[g], adj[g] + np.eye(n))
        s = sizes[g]
        blk = mt[g][:s, :s]                                   # :21-24 vectorised
        blk[blk > 0.0] = 1.0
    return MASK_VALUE * (1.0 - mt)                            # :25


# --------------------------------------------------------------------------
# utils/layers.py:7-46  attn_head (dense additive-mask form)
# --------------------------------------------------------------------------
def attn_head(seq, head, bias_mat, activation=elu, in_drop=0.0, coef_drop=0.0,
              residual=False, return_coef=False, masks=None, res_params=None):
    """Restates utils/layers.py:7-46 for one (meta-path, head) instance.

    seq      (1, N, F)
    head     dict: W (F, F'), a1 (F',), b1 (), a2 (F',), b2 (), c (F',)
             = conv1d/kernel, conv1d_1/{kernel,bias}, conv1d_2/{kernel,bias},
               BiasAdd/biases in creation order.
    bias_mat (1, N, N) additive mask (0 / -1e9)
    masks    None (drop == 0) or dict of {0,1} arrays: 'seq' (1,N,F),
             'coef' (1,N,N), 'fts' (1,N,F') -- the three Bernoulli draws of
             layers.py:19,30,32.
    """
    masks = masks or {}
    seq_in = seq
    if in_drop != 0.0:                                                   # :18-19
        seq = dropout_apply(seq, 1.0 - in_drop, masks.get('seq'))
    seq_fts = seq @ head['W']                                            # :20
    f_1 = seq_fts @ head['a1'][:, None] + head['b1']                     # :23
    f_2 = seq_fts @ head['a2'][:, None] + head['b2']                     # :24
    logits = f_1 + np.transpose(f_2, (0, 2, 1))                          # :26
    coefs = softmax(leaky_relu(logits) + bias_mat, axis=-1)              # :27
    coefs_out = coefs
    if coef_drop != 0.0:                                                 # :29-30
        coefs = dropout_apply(coefs, 1.0 - coef_drop, masks.get('coef'))
    if in_drop != 0.0:                                                   # :31-32
        seq_fts = dropout_apply(seq_fts, 1.0 - in_drop, masks.get('fts'))
    vals = coefs @ seq_fts                                               # :34
    ret = vals + head['c']                                               # :35
    if residual:                                                         # :38-42
        if seq.shape[-1] != ret.shape[-1]:
            # conv1d(seq, F', 1) creates a NEW kernel+bias (use_bias default True)
            ret = ret + (seq @ res_params['W'] + res_params['b'])
        else:
            pass  # :42 assigns to a dead variable -> no-op (bug reproduced)
    del seq_in
    if return_coef:                                                      # :43-46
        # the reference returns the *dropped* coefs tensor (rebinding at :30)
        return activation(ret), (coefs if coef_drop != 0.0 else coefs_out)
    return activation(ret)


def attn_head_const_1(seq, head, bias_mat, activation=elu):
    """utils/layers.py:49-81 (HAN_nd ablation, no dropout): logits := adjacency."""
    adj_mat = 1.0 - bias_mat / MASK_VALUE                                # :56
    seq_fts = seq @ head['W']                                            # :60
    coefs = softmax(leaky_relu(adj_mat) + bias_mat, axis=-1)             # :63-64
    vals = coefs @ seq_fts                                               # :71
    return activation(vals + head['c'])                                  # :72,81


# --------------------------------------------------------------------------
# utils/layers.py:85-127  sp_attn_head (SparseTensor form) on CSR
# --------------------------------------------------------------------------
def sp_attn_head(seq, head, rowptr, colidx, adj_vals=None, activation=elu,
                 in_drop=0.0, coef_drop=0.0, masks=None):
    """Restates utils/layers.py:85-127 with the rank-3 SparseTensor held as CSR.

    logits.values = adj_ij*f1_i + adj_ij*f2_j (:95-96), LeakyReLU on the values
    (:97-99), sparse_softmax over each row's stored entries (:100), dropout on
    the values (:102-106), sparse @ dense (:113-115), bias (:118), activation.
    adj_vals None == binary adjacency.  masks: 'seq' (1,N,F), 'coef' (E,),
    'fts' (1,N,F').
    """
    masks = masks or {}
    n = seq.shape[1]
    if in_drop != 0.0:                                                   # :87-88
        seq = dropout_apply(seq, 1.0 - in_drop, masks.get('seq'))
    seq_fts = (seq @ head['W'])[0]                                       # :90,112
    f_1 = seq_fts @ head['a1'] + head['b1']                              # :93
    f_2 = seq_fts @ head['a2'] + head['b2']                              # :94
    rowptr = np.asarray(rowptr)
    colidx = np.asarray(colidx)
    deg = np.diff(rowptr)
    rows = np.repeat(np.arange(n), deg)
    a = np.ones(colidx.shape[0], dtype=seq_fts.dtype) if adj_vals is None else adj_vals
    lg = leaky_relu(a * f_1[rows] + a * f_2[colidx])                     # :95-99
    fts_used = seq_fts
    if in_drop != 0.0:                                                   # :107-108
        fts_used = dropout_apply(seq_fts[None], 1.0 - in_drop, masks.get('fts'))[0]
    out = np.zeros_like(seq_fts)
    coefs = np.empty_like(lg)
    for i in range(n):                                                   # :100 row softmax
        s, e = rowptr[i], rowptr[i + 1]
        if e > s:
            coefs[s:e] = softmax(lg[s:e])
    cd = coefs
    if coef_drop != 0.0:                                                 # :102-106
        cd = dropout_apply(coefs, 1.0 - coef_drop, masks.get('coef'))
    np.add.at(out, rows, cd[:, None] * fts_used[colidx])                 # :113
    ret = out[None] + head['c']                                          # :114-118
    return activation(ret)


# --------------------------------------------------------------------------
# utils/layers.py:132-164  SimpleAttLayer (semantic-level attention)
# --------------------------------------------------------------------------
def simple_att_layer(inputs, w_omega, b_omega, u_omega, return_alphas=False):
    """Restates utils/layers.py:132-164 (time_major=False).

    inputs (N, P, D); w_omega (D, A); b_omega (A,); u_omega (A,).
    NOTE the softmax is over P *per node* (:157), not the paper's node average.
    """
    v = np.tanh(np.tensordot(inputs, w_omega, axes=1) + b_omega)         # :152
    vu = np.tensordot(v, u_omega, axes=1)                                # :155
    alphas = softmax(vu, axis=-1)                                        # :156
    output = np.sum(inputs * alphas[..., None], axis=1)                  # :159
    if not return_alphas:
        return output
    return output, alphas


# --------------------------------------------------------------------------
# models/gat.py:34-77  HeteGAT_multi.inference
# --------------------------------------------------------------------------
def hetegat_multi_inference(inputs_list, nb_classes, nb_nodes, training, attn_drop,
                            ffd_drop, bias_mat_list, hid_units, n_heads, params,
                            activation=elu, residual=False, mp_att_size=128,
                            masks=None):
    """Restates models/gat.py:34-77.

    params: {'heads': [P][K] head dicts (layer 0),
             'layers': optional [P][len(hid_units)-1][n_heads[i]] head dicts,
             'w_omega','b_omega','u_omega', 'cls': [n_heads[-1]] {'W','b'}}
    masks : optional [P][K] dicts as attn_head's `masks`.
    Returns (logits (1,N,C), final_embed (N, K*F'), att_val (N,P)).
    `training` and `nb_nodes` are accepted and ignored, as in the reference.
    """
    embed_list = []
    for p, (inputs, bias_mat) in enumerate(zip(inputs_list, bias_mat_list)):   # :39
        attns = []
        for k in range(n_heads[0]):                                            # :42-45
            mk = masks[p][k] if masks is not None else None
            attns.append(attn_head(inputs, params['heads'][p][k], bias_mat, activation,
                                   in_drop=ffd_drop, coef_drop=attn_drop,
                                   residual=False, masks=mk))
        h_1 = np.concatenate(attns, axis=-1)                                   # :46
        for i in range(1, len(hid_units)):                                     # :48-57
            attns = []
            for k in range(n_heads[i]):
                lp = params['layers'][p][i - 1][k]
                attns.append(attn_head(h_1, lp, bias_mat, activation,
                                       in_drop=ffd_drop, coef_drop=attn_drop,
                                       residual=residual, res_params=lp.get('res')))
            h_1 = np.concatenate(attns, axis=-1)
        embed_list.append(np.expand_dims(np.squeeze(h_1, axis=0), axis=1))     # :58
    multi_embed = np.concatenate(embed_list, axis=1)                           # :60
    final_embed, att_val = simple_att_layer(multi_embed, params['w_omega'],    # :61-63
                                            params['b_omega'], params['u_omega'],
                                            return_alphas=True)
    out = []
    for i in range(n_heads[-1]):                                               # :66-68
        out.append(final_embed @ params['cls'][i]['W'] + params['cls'][i]['b'])
    logits = sum(out) / n_heads[-1]                                            # :72
    logits = np.expand_dims(logits, axis=0)                                    # :76
    return logits, final_embed, att_val


# --------------------------------------------------------------------------
# models/base_gattn.py:12-24,41-48,61-69  loss / metric / optimiser
# --------------------------------------------------------------------------
def masked_softmax_cross_entropy(logits, labels, mask):
    """models/base_gattn.py:41-48. logits,labels (N,C); mask (N,)."""
    z = logits - np.max(logits, axis=-1, keepdims=True)
    logp = z - np.log(np.sum(np.exp(z), axis=-1, keepdims=True))
    loss = -np.sum(labels * logp, axis=-1)                                     # :43-44
    mask = mask.astype(logits.dtype)                                           # :45
    mask = mask / np.mean(mask)                                                # :46
    return np.mean(loss * mask)                                                # :47-48


def masked_accuracy(logits, labels, mask):
    """models/base_gattn.py:61-69."""
    correct = (np.argmax(logits, 1) == np.argmax(labels, 1)).astype(logits.dtype)
    mask = mask.astype(logits.dtype)
    mask = mask / np.mean(mask)
    return np.mean(correct * mask)


def weighted_loss(logits, labels, nb_classes, class_weights):
    """models/base_gattn.py:5-10 (`loss`): class-weighted sparse softmax cross-entropy, mean over samples.
    logits (N,C); labels (N,) int; class_weights (C,)."""
    z = logits - np.max(logits, axis=-1, keepdims=True)
    logp = z - np.log(np.sum(np.exp(z), axis=-1, keepdims=True))
    sample_wts = np.sum(np.eye(nb_classes)[labels] * class_weights, axis=-1)       # :6-7
    xent = -logp[np.arange(len(labels)), labels] * sample_wts                    # :8-9
    return np.mean(xent)                                                         # :10


def confmat(logits, labels):
    """models/base_gattn.py:33-35: tf.confusion_matrix(labels, argmax(logits)) -- rows = labels,
    columns = predictions, size = max(label, prediction) + 1."""
    preds = np.argmax(logits, axis=1)
    n = int(max(labels.max(), preds.max())) + 1
    cm = np.zeros((n, n), dtype=np.int64)
    np.add.at(cm, (labels, preds), 1)
    return cm


def masked_sigmoid_cross_entropy(logits, labels, mask):
    """models/base_gattn.py:50-59 (multi-label, PPI).  logits, labels (N,C); mask (N,)."""
    x, z = logits, labels.astype(logits.dtype)                                    # :52
    loss = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))              # :53-54 (TF's stable form)
    loss = np.mean(loss, axis=1)                                                 # :55
    mask = mask.astype(logits.dtype)
    mask = mask / np.mean(mask)                                                  # :56-57
    return np.mean(loss * mask)                                                  # :58-59


def micro_f1(logits, labels, mask):
    """models/base_gattn.py:71-94.  round() is round-half-to-even, as tf.round."""
    predicted = np.round(1.0 / (1.0 + np.exp(-logits))).astype(np.int64)         # :73-76
    labels = labels.astype(np.int64)
    m = mask.astype(np.int64)[:, None]                                           # :78-81
    tp = np.count_nonzero(predicted * labels * m)                                # :84
    fp = np.count_nonzero(predicted * (labels - 1) * m)                          # :86
    fn = np.count_nonzero((predicted - 1) * labels * m)                          # :87
    precision = tp / (tp + fp)                                                   # :90
    recall = tp / (tp + fn)                                                      # :91
    return np.float32((2 * precision * recall) / (precision + recall))          # :92-94


def l2_loss_all(param_arrays, l2_coef):
    """models/base_gattn.py:14-16: l2_coef * sum_v sum(v**2)/2 over ALL trainables
    (the name filter never matches a TF variable name such as 'conv1d/bias:0')."""
    return l2_coef * sum(np.sum(np.square(v)) / 2.0 for v in param_arrays)


def adam_step_tf(param, grad, m, v, t, lr=0.005, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (models/base_gattn.py:19-22), step t >= 1:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t * m / (sqrt(v)+eps)."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    param = param - lr_t * m / (np.sqrt(v) + eps)
    return param, m, v


# --------------------------------------------------------------------------
# helpers shared by tests / fixtures (not part of the reference)
# --------------------------------------------------------------------------
def flatten_params(params):
    """Every trainable array in TF creation order (heads per p per k, then
    w_omega, b_omega, u_omega, then classifier kernels/biases)."""
    out = []
    for heads in params['heads']:
        for h in heads:
            out += [h['W'], h['a1'], np.asarray(h['b1']), h['a2'], np.asarray(h['b2']), h['c']]
    out += [params['w_omega'], params['b_omega'], params['u_omega']]
    for c in params['cls']:
        out += [c['W'], c['b']]
    return out


def init_params(rng, n_metapaths, ft_size, nb_classes, hid=8, n_heads=(8, 1),
                mp_att_size=128, dtype=np.float64, nonzero_biases=False, hid_units=None,
                residual=False):
    """Initialisers as the reference's TF defaults.
    nonzero_biases=True perturbs the zero-initialised biases so that parity
    tests exercise them."""
    if hid_units is not None:
        hid = hid_units[0]
    K = n_heads[0]
    D = K * hid

    def bias(shape):
        if nonzero_biases:
            return (0.1 * rng.standard_normal(shape)).astype(dtype)
        return np.zeros(shape, dtype=dtype)

    heads = []
    for _ in range(n_metapaths):
        hp = []
        for _ in range(K):
            hp.append({
                'W': glorot_uniform(rng, ft_size, hid, (ft_size, hid), dtype),
                'a1': glorot_uniform(rng, hid, 1, (hid,), dtype), 'b1': bias(()),
                'a2': glorot_uniform(rng, hid, 1, (hid,), dtype), 'b2': bias(()),
                'c': bias((hid,)),
            })
        heads.append(hp)
    layers = None
    if hid_units is not None and len(hid_units) > 1:     # models/gat.py:48-57
        layers = []
        for _ in range(n_metapaths):
            lp, width = [], D
            for i in range(1, len(hid_units)):
                hi = hid_units[i]
                lp.append([{
                    'W': glorot_uniform(rng, width, hi, (width, hi), dtype),
                    'a1': glorot_uniform(rng, hi, 1, (hi,), dtype), 'b1': bias(()),
                    'a2': glorot_uniform(rng, hi, 1, (hi,), dtype), 'b2': bias(()),
                    'c': bias((hi,)),
                    **({'res': {'W': glorot_uniform(rng, width, hi, (width, hi), dtype), 'b': bias((hi,))}}
                       if (residual and hi != width) else {}),
                } for _ in range(n_heads[i])])
                width = n_heads[i] * hi
            layers.append(lp)
        D = n_heads[len(hid_units) - 1] * hid_units[-1]
    out_extra = {} if layers is None else {'layers': layers}
    return {
        **out_extra,
        'heads': heads,
        'w_omega': (0.1 * rng.standard_normal((D, mp_att_size))).astype(dtype),
        'b_omega': (0.1 * rng.standard_normal((mp_att_size,))).astype(dtype),
        'u_omega': (0.1 * rng.standard_normal((mp_att_size,))).astype(dtype),
        'cls': [{'W': glorot_uniform(rng, D, nb_classes, (D, nb_classes), dtype),
                 'b': bias((nb_classes,))} for _ in range(n_heads[-1])],
    }


def bias_to_csr(bias_mat):
    """(N,N) additive mask -> CSR of the unmasked entries (edge <=> bias > -1e8)."""
    b = np.asarray(bias_mat)
    if b.ndim == 3:
        b = b[0]
    keep = b > -1e8
    rowptr = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int64)
    colidx = np.nonzero(keep)[1].astype(np.int32)
    return rowptr, colidx


def csr_to_bias(rowptr, colidx, n, dtype=np.float64):
    b = np.full((n, n), MASK_VALUE, dtype=dtype)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    b[rows, colidx] = 0.0
    return b[None]
