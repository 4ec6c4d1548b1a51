"""CPU tests of the oracle itself: against the fixtures produced by the one
runnable piece of the reference (adj_to_bias), against the committed golden
forward fixture, and the two independent restatements against each other."""
import os

import numpy as np
import pytest
import torch

from oracle import han_oracle as ho
from oracle import han_oracle_torch as ht
from tests.helpers import make_problem

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases(z):
    return sorted({k[:-4] for k in z.files if k.endswith("_adj")})


def test_adj_to_bias_matches_reference_outputs():
    """tests/golden/adj_to_bias_ref.npz holds outputs of the reference's own
    utils/process.py:14-25 (see tests/golden/gen_fixtures.py)."""
    from han_amd import process
    z = np.load(os.path.join(GOLD, "adj_to_bias_ref.npz"))
    assert len(_cases(z)) >= 5
    for k in _cases(z):
        adj, sizes, nh = z[k + "_adj"], list(z[k + "_sizes"]), int(z[k + "_nhood"])
        want = z[k + "_bias"]
        got_oracle = ho.adj_to_bias(adj.copy(), sizes, nh)
        got_product = process.adj_to_bias(adj.copy(), sizes, nh)
        assert got_oracle.dtype == np.float64 and np.array_equal(got_oracle, want), k
        assert got_product.dtype == np.float64 and np.array_equal(got_product, want), k


def test_adj_to_bias_known_answer():
    """3-node known answer exec'd from the reference lines."""
    a = np.array([[[0., 1., 0.], [1., 0., 0.], [0., 0., 0.]]])
    b = ho.adj_to_bias(a, [3], 1)
    want = np.array([[[0, 0, -1e9], [0, 0, -1e9], [-1e9, -1e9, 0]]], dtype=float)
    assert np.array_equal(np.abs(b), np.abs(want))


def test_adj_to_graph_equals_mask_edges():
    from han_amd import process
    z = np.load(os.path.join(GOLD, "adj_to_bias_ref.npz"))
    for k in ("k3", "n12_h1", "n12_h2", "n10_counts"):
        adj, nh = z[k + "_adj"], int(z[k + "_nhood"])
        rp, ci = ho.bias_to_csr(z[k + "_bias"])
        g = process.adj_to_graph(adj, nhood=nh)
        assert np.array_equal(g.rowptr.numpy(), rp) and np.array_equal(g.colidx.numpy(), ci), k


def test_forward_fixture_reproduces():
    z = np.load(os.path.join(GOLD, "han_forward_n64.npz"))
    P, N = int(z["P"]), int(z["N"])
    bp = {k: torch.tensor(z["param_" + k]) for k in ht.PARAM_ORDER}
    graphs = [(torch.tensor(z[f"rowptr_{p}"]), torch.tensor(z[f"colidx_{p}"])) for p in range(P)]
    x = torch.tensor(z["x"])
    lg, fe, att = ht.hetegat_forward([x] * P, graphs, bp)
    assert np.abs(lg.numpy() - z["logits"]).max() < 1e-12
    assert np.abs(fe.numpy() - z["final_embed"]).max() < 1e-12
    assert np.abs(att.numpy() - z["att_val"]).max() < 1e-12
    assert N == 64


@pytest.mark.parametrize("n,f,p", [(7, 5, 1), (40, 13, 2), (96, 20, 3)])
def test_numpy_and_torch_restatements_agree(n, f, p):
    prob = make_problem(n, n, f, p, 3, [0.1, 0.5, 0.02])
    lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * p, 3, n, False, 0.0, 0.0, prob["biases"],
                                             [8], [8, 1], prob["params"])
    bp = ht.to_batched(prob["params"])
    xt = torch.tensor(prob["x"][0])
    l2, fe2, att2 = ht.hetegat_forward([xt] * p, [torch.tensor(b[0]) for b in prob["biases"]], bp,
                                       dense=True)
    graphs = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    l3, _, _ = ht.hetegat_forward([xt] * p, graphs, bp, dense=False)
    assert np.abs(lg[0] - l2.numpy()).max() < 1e-12
    assert np.abs(fe - fe2.numpy()).max() < 1e-12
    assert np.abs(att - att2.numpy()).max() < 1e-12
    assert np.abs(lg[0] - l3.numpy()).max() < 1e-11       # dense -1e9 mask == neighbours only


def test_dense_mask_equals_sparse_head_fp32():
    """attn_head (layers.py:7-46) == sp_attn_head (layers.py:85-127) on the same
    binary graph, also in the reference's own precision (fp32): the -1e9 mask
    absorbs the logits and exp underflows to exactly 0."""
    prob = make_problem(3, 50, 9, 1, 3, [0.1], dtype=np.float32)
    head = {k: np.asarray(v, dtype=np.float32) for k, v in prob["params"]["heads"][0][0].items()}
    bias = prob["biases"][0].astype(np.float32)
    rp, ci = ho.bias_to_csr(bias)
    d = ho.attn_head(prob["x"], head, bias)
    s = ho.sp_attn_head(prob["x"], head, rp, ci)
    assert d.dtype == np.float32
    assert np.abs(d - s).max() < 2e-6
    _, coefs = ho.attn_head(prob["x"], head, bias, return_coef=True)
    assert np.all(coefs[0][bias[0] < -1e8] == 0.0)


def test_semantic_attention_is_per_node_softmax():
    rng = np.random.default_rng(0)
    m = rng.standard_normal((11, 3, 64))
    w, b, u = rng.standard_normal((64, 128)), rng.standard_normal(128), rng.standard_normal(128)
    out, al = ho.simple_att_layer(m, w, b, u, return_alphas=True)
    assert al.shape == (11, 3) and np.allclose(al.sum(1), 1.0)
    out1, al1 = ho.simple_att_layer(m[4:5], w, b, u, return_alphas=True)
    assert np.allclose(out1, out[4:5]) and np.allclose(al1, al[4:5])     # no cross-node coupling


def test_loss_and_adam_restatements():
    rng = np.random.default_rng(1)
    logits, labels = rng.standard_normal((20, 4)), np.eye(4)[rng.integers(0, 4, 20)]
    mask = rng.random(20) < 0.5
    a = ho.masked_softmax_cross_entropy(logits, labels, mask)
    b = float(ht.masked_softmax_cross_entropy(torch.tensor(logits), torch.tensor(labels), torch.tensor(mask)))
    assert abs(a - b) < 1e-12
    # == mean CE over the masked rows
    z = logits - logits.max(1, keepdims=True)
    ce = -(labels * (z - np.log(np.exp(z).sum(1, keepdims=True)))).sum(1)
    assert abs(a - ce[mask].mean()) < 1e-12
    assert abs(ho.masked_accuracy(logits, labels, mask)
               - (logits.argmax(1) == labels.argmax(1))[mask].mean()) < 1e-12
    p, m, v = np.ones(3), np.zeros(3), np.zeros(3)
    p, m, v = ho.adam_step_tf(p, np.array([1.0, -2.0, 0.0]), m, v, 1)
    # first TF-Adam step moves by ~lr in the direction of -sign(g)
    assert np.allclose(p[:2], [1 - 0.005, 1 + 0.005], atol=1e-6) and p[2] == 1.0


def test_dropout_semantics():
    x = np.ones((4, 5))
    mask = np.zeros((4, 5)); mask[0] = 1
    y = ho.dropout_apply(x, 0.4, mask)
    assert np.allclose(y[0], 2.5) and np.all(y[1:] == 0)
    assert ho.dropout_apply(x, 1.0, None) is x
