"""Shared builders for the parity tests (test infrastructure)."""
import numpy as np
import torch

from oracle import han_oracle as ho
from oracle import han_oracle_torch as ht


def random_adj(rng, n, density, symmetric=True, special_rows=True, nnz=None):
    """Binary adjacency WITHOUT self-loops (adj_to_bias re-adds I, as
    ex_acm3025.py:61 / utils/process.py:18-20).  `density` = the probability of every off-diagonal pair
    (the strict upper triangle is drawn and mirrored; round 3 drew the full matrix and took max(a, a^T), i.e.
    2d - d^2); nnz = EXACT number of entries the graph has after +I (the data sets' counts include the
    self-loops), in which case no special rows are planted.  special_rows plants the edge
    cases of SURVEY.md section 4: an isolated node (degree 1 after +I: the
    self-loop only) and a node adjacent to everybody (degree N)."""
    if symmetric:
        iu, ju = np.triu_indices(n, 1)
        if nnz is not None:
            assert (nnz - n) % 2 == 0
            sel = rng.choice(iu.size, size=(nnz - n) // 2, replace=False)
            special_rows = False
        else:
            sel = np.nonzero(rng.random(iu.size) < density)[0]
        a = np.zeros((n, n), dtype=np.float64)
        a[iu[sel], ju[sel]] = 1.0
        a[ju[sel], iu[sel]] = 1.0
    else:
        a = (rng.random((n, n)) < density).astype(np.float64)
    np.fill_diagonal(a, 0.0)
    if special_rows and n >= 4:
        a[1, :] = 0.0
        a[:, 1] = 0.0          # isolated node
        a[2, :] = 1.0
        a[:, 2] = 1.0          # hub
        a[2, 2] = 0.0
        a[1, 2] = a[2, 1] = 0.0
    return a


def make_problem(seed, n, f, p, c, densities, dtype=np.float64, nonzero_biases=True, hid_units=None,
                 n_heads=(8, 1), residual=False, mp_att_size=128, nnz=None):
    """nnz: per meta-path the exact entry count incl. self-loops (a data set's figure) instead of a density."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((1, n, f)).astype(dtype)
    if nnz is not None:
        adjs = [random_adj(rng, n, None, nnz=nnz[i % len(nnz)])[None] for i in range(p)]
    else:
        adjs = [random_adj(rng, n, densities[i % len(densities)])[None] for i in range(p)]
    biases = [ho.adj_to_bias(a, [n], 1) for a in adjs]
    params = ho.init_params(rng, p, f, c, nonzero_biases=nonzero_biases, hid_units=hid_units,
                            n_heads=n_heads, residual=residual, mp_att_size=mp_att_size)
    labels = rng.integers(0, c, size=n)
    onehot = np.eye(c)[labels]
    mask = rng.random(n) < 0.4
    mask[0] = True
    return dict(x=x, adjs=adjs, biases=biases, params=params, labels=labels, onehot=onehot,
                mask=mask, n=n, f=f, p=p, c=c)


def load_params(model, bp):
    """Copy batched oracle parameters (han_oracle_torch.to_batched) into a built model."""
    with torch.no_grad():
        for k in ht.param_order(bp):
            getattr(model, k).copy_(bp[k].to(torch.float32))


def build_model(prob, dev, mp_att_size=None):
    from han_amd.gat import HeteGAT_multi
    model = HeteGAT_multi()
    if mp_att_size is None:
        mp_att_size = int(np.asarray(prob["params"]["w_omega"]).shape[1])
    heads0 = prob["params"]["heads"][0]
    k0, fp0 = len(heads0), len(heads0[0]["a1"])          # first layer: K heads of width F'
    hid_units, n_heads = (fp0,), (k0, len(prob["params"]["cls"]))
    if "layers" in prob["params"]:
        lp = prob["params"]["layers"][0]
        hid_units = (fp0,) + tuple(len(l[0]["a1"]) for l in lp)
        n_heads = (k0,) + tuple(len(l) for l in lp) + (len(prob["params"]["cls"]),)
    residual = "layers" in prob["params"] and "res" in prob["params"]["layers"][0][0][0]
    model.build(prob["p"], prob["f"], prob["c"], hid_units, n_heads, mp_att_size, device=dev,
                residual=residual)
    bp = ht.to_batched(prob["params"])
    load_params(model, bp)
    return model, bp


def gpu_inputs(prob, dev):
    from han_amd.graph import CSRGraph
    x = torch.tensor(prob["x"][0], dtype=torch.float32, device=dev)
    graphs = []
    for b in prob["biases"]:
        rp, ci = ho.bias_to_csr(b)
        graphs.append(CSRGraph.from_arrays(rp, ci, prob["n"], device=dev))
    return x, graphs


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def group_masks(seed, n, f, K, FP, rowptr, colidx, drop, row_offset=0):
    """The dropout masks the kernels draw for a layer of K heads of width FP (han_amd.gat.node_level): a
    head runs at the kernel width FPk = next of 4/8/16/32/64 (zero-weight columns beyond FP), the heads in
    groups of 64 // FPk (64 columns per K1/K2 launch); group g uses seed + g, a head's RNG key is its index
    INSIDE the group and a column's key its position in the 64-column group layout.
    Returns the oracle's mask dict: seq (K,N,F), coef (E,K), fts (N,K*FP)."""
    import torch
    from tests import rng_ref
    if FP > 64:       # a head wider than 64 columns: one head per group (seed + k, head index 0), slices of 64 columns
        S = -(-FP // 64)       # that share the head's draws except for the per-column projected-row dropout (stream 2 + 4 s)
        seq, coef, fts = [], [], []
        for k in range(K):
            sd = (int(seed) + k) & ((1 << 64) - 1)
            seq.append(rng_ref.seq_mask(sd, n, f, 1, drop, row_offset))
            coef.append(rng_ref.coef_mask_csr(sd, rowptr, colidx, 1, drop, row_offset))
            fts.append(np.concatenate([rng_ref.fts_mask(sd, n, 64, drop, row_offset, slice_index=s_)
                                       for s_ in range(S)], 1)[:, :FP])
        return {"seq": torch.tensor(np.concatenate(seq, 0)), "coef": torch.tensor(np.concatenate(coef, 1)),
                "fts": torch.tensor(np.concatenate(fts, 1))}
    FPk = next(w for w in (4, 8, 16, 32, 64) if FP <= w)
    kg = 64 // FPk
    seq, coef, fts = [], [], []
    for g0 in range(0, K, kg):
        nh = min(kg, K - g0)
        sd = (int(seed) + g0 // kg) & ((1 << 64) - 1)
        seq.append(rng_ref.seq_mask(sd, n, f, kg, drop, row_offset)[:nh])
        coef.append(rng_ref.coef_mask_csr(sd, rowptr, colidx, kg, drop, row_offset)[:, :nh])
        m = rng_ref.fts_mask(sd, n, 64, drop, row_offset).reshape(n, kg, FPk)[:, :nh, :FP]
        fts.append(m.reshape(n, nh * FP))
    return {"seq": torch.tensor(np.concatenate(seq, 0)), "coef": torch.tensor(np.concatenate(coef, 1)),
            "fts": torch.tensor(np.concatenate(fts, 1))}
