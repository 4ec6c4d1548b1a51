"""GPU parity tests: every HIP kernel, through the C ABI (han_amd.ops -> ctypes ->
libhan_hip.so), against the CPU oracle on the same seeded inputs.

Tolerances (fp32 kernels vs float64 oracle): 1e-4 absolute on logits / outputs
(the bar BASELINE.json's north_star states), 2e-3 relative-to-max on gradients.
PARITY UNPINNED with respect to TensorFlow itself (oracle/han_oracle.py header).
"""
import numpy as np
import pytest
import torch

from oracle import han_oracle as ho
from oracle import han_oracle_torch as ht
from tests import rng_ref
from tests.helpers import load_params, build_model, gpu_inputs, group_masks, make_problem, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4
GTOL = 2e-3


def _t(a, dev, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device=dev)


# ----------------------------------------------------------------------------- K1
@pytest.mark.parametrize("n,f", [(1, 5), (7, 13), (64, 32), (130, 77), (512, 334), (3025, 1870), (700, 129)])
def test_project_fwd_matches_oracle(dev, n, f):
    from han_amd import ops
    rng = np.random.default_rng(n * 1000 + f)
    x = rng.standard_normal((n, f))
    W = rng.standard_normal((f, 64)) * 0.2
    a1, a2 = rng.standard_normal((8, 8)), rng.standard_normal((8, 8))
    b1, b2 = rng.standard_normal(8), rng.standard_normal(8)
    H, f1, f2 = ops.project_fwd(_t(x, dev), _t(W, dev), _t(a1, dev), _t(a2, dev), _t(b1, dev),
                                       _t(b2, dev))
    Href = x @ W
    assert np.abs(H.cpu().numpy() - Href).max() < TOL * max(1.0, np.abs(Href).max())
    f1ref = (Href.reshape(n, 8, 8) * a1[None]).sum(-1) + b1
    f2ref = (Href.reshape(n, 8, 8) * a2[None]).sum(-1) + b2
    assert np.abs(f1.cpu().numpy() - f1ref).max() < 1e-3 * max(1.0, np.abs(f1ref).max())
    assert np.abs(f2.cpu().numpy() - f2ref).max() < 1e-3 * max(1.0, np.abs(f2ref).max())


@pytest.mark.parametrize("n,f", [(7, 13), (130, 77), (130, 300)])      # (130, 300): split-F path
def test_project_dropout_matches_hash_masks(dev, n, f):
    """Input dropout re-sampled per head (layers.py:18-19) and projected-row
    dropout (layers.py:31-32) with the masks regenerated in NumPy."""
    from han_amd import ops
    rng = np.random.default_rng(5)
    seed, drop, off = 0x1234ABCD5678, 0.6, 11
    x = rng.standard_normal((n, f))
    W = rng.standard_normal((f, 64)) * 0.2
    a1, a2 = rng.standard_normal((8, 8)), rng.standard_normal((8, 8))
    b1, b2 = rng.standard_normal(8), rng.standard_normal(8)
    H, f1, _ = ops.project_fwd(_t(x, dev), _t(W, dev), _t(a1, dev), _t(a2, dev), _t(b1, dev),
                                      _t(b2, dev), in_drop=drop, fts_drop=drop, seed=seed, row_offset=off)
    keep = rng_ref.keep_prob32(drop)
    sm = rng_ref.seq_mask(seed, n, f, 8, drop, row_offset=off)
    Href = np.concatenate([(x / keep * sm[k]) @ W[:, 8 * k:8 * k + 8] for k in range(8)], 1)
    assert np.abs(H.cpu().numpy() - Href).max() < TOL * max(1.0, np.abs(Href).max())
    fm = rng_ref.fts_mask(seed, n, 64, drop, row_offset=off)
    bits = H.cpu().numpy().view(np.uint32) & 1                   # keep bits ride in mantissa bit 0
    assert np.array_equal(bits.astype(np.float64), fm)
    # backward: dW = sum_k masked X^T dH_k
    dH = rng.standard_normal((n, 64))
    dW = ops.project_bwd(_t(x, dev), _t(dH, dev), 8, 8, in_drop=drop, seed=seed, row_offset=off)
    dWref = np.concatenate([(x / keep * sm[k]).T @ dH[:, 8 * k:8 * k + 8] for k in range(8)], 1)
    assert rel_err(dW.cpu().numpy(), dWref) < 1e-4


@pytest.mark.parametrize("variant", ["1"])
@pytest.mark.parametrize("n,f,xbf", [(16384 + 77, 64, False), (20000, 1870 - 2, False), (17000, 256, True),
                                     (40000, 36, False), (16500, 200, True)])
def test_project_matrix_pipe_kernel_fp32_class_accuracy(dev, n, f, xbf, variant, monkeypatch):
    """Large inputs run K1 on the bf16 matrix pipe with an exact three-way bf16 split of both operands
    (six products; project_fwd_b6_kernel).  It must be as accurate as the exact-fp32 MFMA kernel: both
    against the float64 product, at the ACM feature width (1868 ~ 1870, F % 4 == 0) too; F = 36 has a
    partial last K-step; xbf: bf16 features (their split is one term)."""
    from han_amd import ops
    rng = np.random.default_rng(n + f)
    x = rng.standard_normal((n, f)) * np.exp(rng.standard_normal((n, 1)))       # rows of very different scale
    W = rng.standard_normal((f, 64)) * 0.2
    a = [rng.standard_normal((8, 8)) for _ in range(2)] + [rng.standard_normal(8) for _ in range(2)]
    xt = _t(x, dev)
    if xbf:
        xt = xt.to(torch.bfloat16)
        x = xt.to(torch.float32).cpu().numpy().astype(np.float64)
    args = (xt, _t(W, dev), _t(a[0], dev), _t(a[1], dev), _t(a[2], dev), _t(a[3], dev))
    Href = x @ W.astype(np.float32).astype(np.float64)
    scale = np.abs(x) @ np.abs(W)                    # sum |x w|: what an fp32 dot product's error scales with
    H32, _, _ = ops.project_fwd(*args, flags=ops.FLAG_K1_EXACT_PIPE)
    H6, f1, f2 = ops.project_fwd(*args, flags=ops.FLAG_K1_MATRIX_PIPE)
    e32 = np.abs(H32.cpu().numpy() - Href) / scale
    e6 = np.abs(H6.cpu().numpy() - Href) / scale
    assert e6.max() < 4e-7, (e6.max(), e32.max())      # a few fp32 ulps of sum |x w|, like the fp32 pipe
    assert e6.max() < 4 * max(e32.max(), 5e-8)
    f1ref = (Href.reshape(n, 8, 8) * a[0][None]).sum(-1) + a[2]
    assert np.abs(f1.cpu().numpy() - f1ref).max() < 1e-3 * max(1.0, np.abs(f1ref).max())


@pytest.mark.parametrize("variant", ["1"])
@pytest.mark.parametrize("f,xbf", [(44, False), (256, False), (64, True), (328, False)])
def test_project_matrix_pipe_kernel_dropout_masks(dev, f, xbf, variant, monkeypatch):
    """The matrix-pipe kernel with the per-head input dropout (layers.py:18-19: packed 16-bit keep masks on
    the A fragments) and the projected-row keep bits, against the NumPy-regenerated hash masks -- and
    bit-for-bit the same keep decisions as the fp32 kernel."""
    from han_amd import ops
    n = 16384 + 130
    rng = np.random.default_rng(f)
    seed, drop, off = 0x0BADC0DE1234, 0.6, 977
    x = rng.standard_normal((n, f))
    W = rng.standard_normal((f, 64)) * 0.2
    a = [rng.standard_normal((8, 8)) for _ in range(2)] + [rng.standard_normal(8) for _ in range(2)]
    xt = _t(x, dev)
    if xbf:
        xt = xt.to(torch.bfloat16)
        x = xt.to(torch.float32).cpu().numpy().astype(np.float64)
    H, _, _ = ops.project_fwd(xt, _t(W, dev), _t(a[0], dev), _t(a[1], dev), _t(a[2], dev), _t(a[3], dev),
                              in_drop=drop, fts_drop=drop, seed=seed, row_offset=off)
    keep = rng_ref.keep_prob32(drop)
    sm = rng_ref.seq_mask(seed, n, f, 8, drop, row_offset=off)
    Wd = W.astype(np.float32).astype(np.float64)
    Href = np.concatenate([(x / keep * sm[k]) @ Wd[:, 8 * k:8 * k + 8] for k in range(8)], 1)
    assert np.abs(H.cpu().numpy() - Href).max() < 1e-5 * max(1.0, np.abs(Href).max())
    fm = rng_ref.fts_mask(seed, n, 64, drop, row_offset=off)
    assert np.array_equal((H.cpu().numpy().view(np.uint32) & 1).astype(np.float64), fm)
    # the backward (exact-fp32 dW kernel) regenerates the very same draws
    dH = rng.standard_normal((n, 64))
    dW = ops.project_bwd(xt, _t(dH, dev), 8, 8, in_drop=drop, seed=seed, row_offset=off)
    dWref = np.concatenate([(x / keep * sm[k]).T @ dH[:, 8 * k:8 * k + 8] for k in range(8)], 1)
    assert rel_err(dW.cpu().numpy(), dWref) < 2e-6
    # other drop rates exercise the threshold arithmetic of the packed compare (odd / even thresholds)
    for dr in (0.25, 0.9):
        H2, _, _ = ops.project_fwd(xt, _t(W, dev), _t(a[0], dev), _t(a[1], dev), _t(a[2], dev), _t(a[3], dev),
                                   in_drop=dr, seed=seed, row_offset=off)
        kp = rng_ref.keep_prob32(dr)
        sm2 = rng_ref.seq_mask(seed, n, f, 8, dr, row_offset=off)
        Href2 = np.concatenate([(x / kp * sm2[k]) @ Wd[:, 8 * k:8 * k + 8] for k in range(8)], 1)
        assert np.abs(H2.cpu().numpy() - Href2).max() < 1e-5 * max(1.0, np.abs(Href2).max()), dr


@pytest.mark.parametrize("n,f", [(33, 70), (3025, 130)])
def test_project_bwd_no_dropout(dev, n, f):
    from han_amd import ops
    rng = np.random.default_rng(9)
    x, dH = rng.standard_normal((n, f)), rng.standard_normal((n, 64))
    dW = ops.project_bwd(_t(x, dev), _t(dH, dev), 8, 8)
    assert rel_err(dW.cpu().numpy(), x.T @ dH) < 1e-4


@pytest.mark.parametrize("n,f,xbf", [(32768 + 77, 256, False), (40000, 72, False), (33000, 264, True)])
def test_project_keep_table_and_block_mfma_dw(dev, n, f, xbf):
    """Round 3: the training forward writes the keep table of the per-head input dropout (one 64-bit word per
    row and feature octet, bit order of han_hip.h) and dW reads it as the lane mask of its 4x4x1 16-block MFMA
    A operand.  The table must hold exactly the regenerated hash draws, dW through it must equal the float64
    product with those masks AND the hash-regenerating dW kernel (same draws, other kernel)."""
    from han_amd import ops
    rng = np.random.default_rng(n + f)
    seed, drop, off = 0x5EED0000 + f, 0.6, 4242
    x = rng.standard_normal((n, f))
    W = rng.standard_normal((f, 64)) * 0.2
    a = [rng.standard_normal((8, 8)) for _ in range(2)] + [rng.standard_normal(8) for _ in range(2)]
    xt = _t(x, dev)
    if xbf:
        xt = xt.to(torch.bfloat16)
        x = xt.to(torch.float32).cpu().numpy().astype(np.float64)
    H, f1, f2, keep = ops.project_fwd(xt, _t(W, dev), _t(a[0], dev), _t(a[1], dev), _t(a[2], dev), _t(a[3], dev),
                                      in_drop=drop, fts_drop=drop, seed=seed, row_offset=off, want_keep=True)
    assert keep is not None and keep.numel() == n * f + 128
    H0, _, _ = ops.project_fwd(xt, _t(W, dev), _t(a[0], dev), _t(a[1], dev), _t(a[2], dev), _t(a[3], dev),
                               in_drop=drop, fts_drop=drop, seed=seed, row_offset=off)
    assert torch.equal(H, H0)                                    # writing the table changes nothing else
    sm = rng_ref.seq_mask(seed, n, f, 8, drop, row_offset=off)   # (8, n, f) 0/1
    words = keep[:n * f].cpu().numpy().view("<u8").reshape(n, f // 8)
    for k in range(8):
        for q in range(2):
            for i in range(4):
                bit = 32 * q + 16 * (k % 2) + 4 * (k // 2) + i
                got = ((words >> np.uint64(bit)) & np.uint64(1)).astype(np.float64)
                assert np.array_equal(got, sm[k][:, 4 * q + i::8]), (k, q, i)
    kp = rng_ref.keep_prob32(drop)
    dH = rng.standard_normal((n, 64))
    dW = ops.project_bwd(xt, _t(dH, dev), 8, 8, in_drop=drop, seed=seed, row_offset=off, keep=keep)
    dWref = np.concatenate([(x / kp * sm[k]).T @ dH[:, 8 * k:8 * k + 8] for k in range(8)], 1)
    assert rel_err(dW.cpu().numpy(), dWref) < 2e-6
    dWh = ops.project_bwd(xt, _t(dH, dev), 8, 8, in_drop=drop, seed=seed, row_offset=off)
    assert rel_err(dW.cpu().numpy(), dWh.cpu().numpy().astype(np.float64)) < 2e-6
    # no table for small inputs / other head shapes: the wrapper returns None and dW regenerates the draws
    assert ops.project_fwd(xt[:5000], _t(W, dev), _t(a[0], dev), _t(a[1], dev), _t(a[2], dev), _t(a[3], dev),
                           in_drop=drop, seed=seed, want_keep=True)[3] is None


@pytest.mark.parametrize("P,n,f,xbf,K,FP,tdt", [(4, 16384 + 200, 256, False, 8, 8, torch.float32),
                                                (3, 17000, 72, False, 8, 8, torch.float32),
                                                (2, 20000, 132, True, 8, 8, torch.bfloat16),
                                                (5, 16500, 64, False, 4, 16, torch.float32),
                                                (2, 3000, 256, False, 8, 8, torch.float32)])
@pytest.mark.parametrize("flags", [0, 32])
def test_project_fwd_multi_matches_the_per_meta_path_kernel(dev, P, n, f, xbf, K, FP, tdt, flags):
    """Round 3: the eval forward of all P meta-paths of a shared X in one fused launch (X read, split and staged once
    for 4 or 2 meta-paths per block; odd P leaves one to the single kernel; short inputs and other cases loop):
    H against the float64 product, and H / f1 / f2 against the per-meta-path kernel on the same matrix pipe."""
    from han_amd import ops
    rng = np.random.default_rng(P * 100 + f)
    x = rng.standard_normal((n, f)) * np.exp(0.5 * rng.standard_normal((n, 1)))
    W = rng.standard_normal((P, f, 64)) * 0.2
    a1, a2 = rng.standard_normal((P, K, FP)), rng.standard_normal((P, K, FP))
    b1, b2 = rng.standard_normal((P, K)), rng.standard_normal((P, K))
    xt = _t(x, dev).to(torch.bfloat16) if xbf else _t(x, dev)
    xf = xt.to(torch.float32).cpu().numpy().astype(np.float64)
    Wt, a1t, a2t, b1t, b2t = (_t(v, dev) for v in (W, a1, a2, b1, b2))
    H, f1, f2 = ops.project_fwd_multi(xt, Wt, a1t, a2t, b1t, b2t, table_dtype=tdt, flags=flags)
    assert H.shape == (P, n, 64) and f1.shape == (P, n, K)
    for p in range(P):
        Href = xf @ W[p].astype(np.float32).astype(np.float64)
        Hp = H[p].to(torch.float32).cpu().numpy()
        tol = 1e-2 if tdt == torch.bfloat16 else 2e-6
        assert np.abs(Hp - Href).max() < tol * max(1.0, np.abs(Href).max()), p
        Hs, g1, g2 = ops.project_fwd(xt, Wt[p], a1t[p], a2t[p], b1t[p], b2t[p], table_dtype=tdt,
                                     flags=ops.FLAG_K1_MATRIX_PIPE)
        if n >= 16384 and p < P - P % 2:      # fused: the same six products in the same order, bit-identical rows
            assert torch.equal(H[p], Hs), p      # (an odd last meta-path runs the single-path kernel)
        st = Hp.astype(np.float64).reshape(n, K, FP)
        f1ref = (st * a1[p][None]).sum(-1) + b1[p]
        f2ref = (st * a2[p][None]).sum(-1) + b2[p]
        assert np.abs(f1[p].cpu().numpy() - f1ref).max() < 1e-4 * max(1.0, np.abs(f1ref).max()), p
        assert np.abs(f2[p].cpu().numpy() - f2ref).max() < 1e-4 * max(1.0, np.abs(f2ref).max()), p


def test_bf16_table_keeps_nan_and_inf(dev):
    """han_f32_to_bf16_bits (VERDICT r3 hygiene): a NaN stays a NaN in a bf16 table -- the rounding add used to carry a
    NaN whose upper mantissa bits were all ones into the sign / exponent (0x7FFFFFFF -> -0.0) -- and +-inf stay +-inf."""
    from han_amd import ops
    n, f = 300, 8
    X = torch.zeros((n, f), device=dev)
    X[:, 0] = 1.0
    X[7, 0] = float("nan")
    X[8, 0] = torch.tensor(0x7FFFFFFF, dtype=torch.int32).view(torch.float32)      # the NaN that used to turn into -0.0
    X[9, 0] = float("inf")
    X[10, 0] = -float("inf")
    W = torch.zeros((f, 64), device=dev)
    W[0, :] = 1.0                                  # H[n, :] = X[n, 0]
    a = torch.zeros((8, 8), device=dev)
    b = torch.zeros(8, device=dev)
    H, f1, f2 = ops.project_fwd(X, W, a, a, b, b, table_dtype=torch.bfloat16)
    H = H.float()
    assert torch.isnan(H[7]).all() and torch.isnan(H[8]).all()
    assert (H[9] == float("inf")).all() and (H[10] == -float("inf")).all()
    assert (H[11] == 1.0).all()


@pytest.mark.parametrize("K,FP", [(4, 16), (16, 4), (2, 32)])
@pytest.mark.parametrize("n,f,xbf", [(3000, 256, False), (16500, 64, True)])
def test_bf16_table_scores_follow_the_head_width(dev, K, FP, n, f, xbf):
    """with a bf16 H table the separate scores launch (split-F path: few rows, F >= 128; and
    the matrix-pipe eval forward: bf16 features, N >= 16384) used the 8 x 8 lane map for every head shape --
    out-of-bounds f1 / f2 writes for K < 8, a wrong layout for K = 16.  f1 / f2 against the rows as stored."""
    from han_amd import ops
    rng = np.random.default_rng(K * 1000 + n)
    x = rng.standard_normal((n, f))
    W = rng.standard_normal((f, 64)) * 0.2
    a1, a2, b1, b2 = rng.standard_normal((K, FP)), rng.standard_normal((K, FP)), rng.standard_normal(K), rng.standard_normal(K)
    xt = _t(x, dev).to(torch.bfloat16) if xbf else _t(x, dev)
    H, f1, f2 = ops.project_fwd(xt, _t(W, dev), _t(a1, dev), _t(a2, dev), _t(b1, dev), _t(b2, dev),
                                table_dtype=torch.bfloat16)
    Hs = H.to(torch.float32).cpu().numpy().astype(np.float64).reshape(n, K, FP)
    f1ref = (Hs * a1[None]).sum(-1) + b1
    f2ref = (Hs * a2[None]).sum(-1) + b2
    assert f1.shape == (n, K) and np.abs(f1.cpu().numpy() - f1ref).max() < 1e-4 * max(1.0, np.abs(f1ref).max())
    assert np.abs(f2.cpu().numpy() - f2ref).max() < 1e-4 * max(1.0, np.abs(f2ref).max())
    xf = xt.to(torch.float32).cpu().numpy().astype(np.float64)
    Href = xf @ W.astype(np.float32).astype(np.float64)
    assert np.abs(H.to(torch.float32).cpu().numpy() - Href).max() < 1e-2 * max(1.0, np.abs(Href).max())


# ----------------------------------------------------------------------------- K2
def _k2_inputs(rng, n, density, dev):
    from han_amd.graph import CSRGraph
    from tests.helpers import random_adj
    adj = random_adj(rng, n, density)
    bias = ho.adj_to_bias(adj[None], [n], 1)
    rp, ci = ho.bias_to_csr(bias)
    H = rng.standard_normal((n, 64))
    f1 = rng.standard_normal((n, 8)) * 2
    a2, b2 = rng.standard_normal((8, 8)), rng.standard_normal(8)
    c = rng.standard_normal(64) * 0.1
    g = CSRGraph.from_arrays(rp, ci, n, device=dev)
    return bias, rp, ci, H, f1, a2, b2, c, g


def _f2(H, a2, b2):
    """layers.py:24: f_2 = conv1d(seq_fts, 1, 1) per head."""
    return (H.reshape(H.shape[0], 8, 8) * a2[None]).sum(-1) + b2


def _k2_oracle(bias, H, f1, f2, c):
    """layers.py:26-35,46 per head from given H, f1, f2 (dense additive mask)."""
    n = H.shape[0]
    out = np.zeros((n, 64))
    lse = np.zeros((n, 8))
    for k in range(8):
        logits = ho.leaky_relu(f1[:, k][:, None] + f2[:, k][None, :]) + bias[0]
        coefs = ho.softmax(logits, axis=-1)
        out[:, 8 * k:8 * k + 8] = coefs @ H[:, 8 * k:8 * k + 8] + c[8 * k:8 * k + 8]
        mx = logits.max(1)
        lse[:, k] = mx + np.log(np.exp(logits - mx[:, None]).sum(1))
    return ho.elu(out), out, lse


@pytest.mark.parametrize("n,density", [(7, 0.3), (64, 0.05), (64, 0.5), (512, 0.01), (512, 0.3),
                                        (1000, 0.002)])
def test_node_attn_fwd_matches_dense_mask_oracle(dev, n, density):
    """CSR gather kernel == the reference's dense -1e9 additive-mask softmax,
    incl. a self-loop-only row, a full row, low- and high-degree variants."""
    from han_amd import ops
    rng = np.random.default_rng(int(n * 100 + density * 1000))
    bias, rp, ci, H, f1, a2, b2, c, g = _k2_inputs(rng, n, density, dev)
    ref, pre_ref, lse_ref = _k2_oracle(bias, H, f1, _f2(H, a2, b2), c)
    args = (g, _t(H, dev), _t(f1, dev), _t(a2, dev), _t(b2, dev), _t(c, dev))
    out, saved = ops.node_attn_fwd(*args)
    assert saved is None
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    out2, saved = ops.node_attn_fwd(*args, train=True)
    kept, lse, aggp, tsum = saved
    assert np.abs(out2.cpu().numpy() - ref).max() < TOL
    # round 3: the pre-activation is not stored; saved[0] is the output itself, from which the backward recovers it
    assert kept.data_ptr() == out2.data_ptr()
    inv = np.where(out2.cpu().numpy() > 0, out2.cpu().numpy(), np.log1p(np.minimum(out2.cpu().numpy().astype(np.float64), 0)))
    assert np.abs(inv - pre_ref).max() < 10 * TOL
    assert np.abs(lse.cpu().numpy() - lse_ref).max() < 1e-4 * max(1.0, np.abs(lse_ref).max())
    # strided output straight into M[:, p, :]
    M = torch.zeros((n, 3, 64), device=dev)
    ops.node_attn_fwd(*args, out=M[:, 1, :])
    assert np.abs(M[:, 1, :].cpu().numpy() - ref).max() < TOL
    assert float(M[:, 0, :].abs().max()) == 0.0 and float(M[:, 2, :].abs().max()) == 0.0


def test_node_attn_online_softmax_rescale(dev):
    """Forces the running-max rescale: the largest score arrives LAST in a long
    row, after hundreds of small ones, with a spread > 80 (exp underflow range)."""
    from han_amd import ops
    from han_amd.graph import CSRGraph
    n = 700
    rng = np.random.default_rng(3)
    H = rng.standard_normal((n, 64))
    f1 = np.zeros((n, 8))
    a2 = np.zeros((8, 8)); a2[:, 0] = 1.0      # f2[j,k] = H[j, 8k]
    b2 = np.zeros(8)
    H[-1, ::8] = 90.0        # last neighbour of every row dominates
    H[-2, ::8] = -90.0
    f2 = _f2(H, a2, b2)
    rp = np.arange(0, n * n + 1, n)
    ci = np.tile(np.arange(n), n).astype(np.int32)
    bias = np.zeros((1, n, n))
    c = np.zeros(64)
    ref, _, _ = _k2_oracle(bias, H, f1, f2, c)
    g = CSRGraph.from_arrays(rp, ci, n, device=dev)
    out, _ = ops.node_attn_fwd(g, _t(H, dev), _t(f1, dev), _t(a2, dev), _t(b2, dev), _t(c, dev))
    assert np.abs(out.cpu().numpy() - ref).max() < TOL


def test_node_attn_empty_rows_and_determinism(dev):
    from han_amd import ops
    from han_amd.graph import CSRGraph
    n = 9
    rp = np.array([0, 0, 2, 2, 5, 5, 5, 6, 6, 6])
    ci = np.array([0, 3, 1, 2, 8, 7], dtype=np.int32)
    rng = np.random.default_rng(0)
    H, f1 = rng.standard_normal((n, 64)), rng.standard_normal((n, 8))
    a2, b2 = rng.standard_normal((8, 8)), rng.standard_normal(8)
    c = rng.standard_normal(64)
    g = CSRGraph.from_arrays(rp, ci, n, device=dev)
    out, _ = ops.node_attn_fwd(g, _t(H, dev), _t(f1, dev), _t(a2, dev), _t(b2, dev), _t(c, dev))
    o = out.cpu().numpy()
    assert np.isfinite(o).all()
    # a row with no stored entry aggregates nothing: out = act(0 + c) (sp_attn_head, layers.py:113-118)
    assert np.abs(o[0] - ho.elu(c)).max() < 1e-6
    out_b, _ = ops.node_attn_fwd(g, _t(H, dev), _t(f1, dev), _t(a2, dev), _t(b2, dev), _t(c, dev))
    assert torch.equal(out, out_b)      # no float atomics: bitwise reproducible


# ------------------------------------------------------------------- full forward
@pytest.mark.parametrize("n,f,p,dens", [(7, 5, 1, [0.4]), (64, 20, 2, [0.1, 0.6]),
                                         (512, 48, 3, [0.004, 0.05, 0.4])])
def test_inference_matches_oracle(dev, n, f, p, dens):
    """HeteGAT_multi.inference == oracle restatement of models/gat.py:34-77
    (dense bias_mat path), logits within 1e-4."""
    prob = make_problem(100 + n, n, f, p, 3, dens)
    lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * p, 3, n, False, 0.0, 0.0, prob["biases"],
                                             [8], [8, 1], prob["params"])
    model, _ = build_model(prob, dev)
    x = _t(prob["x"], dev)                                    # (1,N,F) as the reference feeds
    biases = [_t(b, dev) for b in prob["biases"]]             # dense (1,N,N) masks -> GPU CSR builder
    with torch.no_grad():
        logits, final_embed, att_val = model.inference([x] * (p + 1), 3, n, False, 0.0, 0.0, biases,
                                                       [8], [8, 1])
    assert logits.shape == (1, n, 3) and final_embed.shape == (n, 64) and att_val.shape == (n, p)
    assert np.abs(logits.cpu().numpy() - lg).max() < TOL
    assert np.abs(final_embed.cpu().numpy() - fe).max() < TOL
    assert np.abs(att_val.cpu().numpy() - att).max() < TOL


def test_inference_matches_golden_fixture(dev):
    """Committed fixture (tests/golden/han_forward_n64.npz, generated by
    tests/golden/gen_fixtures.py from the float64 oracle)."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "han_forward_n64.npz")
    z = np.load(path)
    from han_amd.gat import HeteGAT_multi
    from han_amd.graph import CSRGraph
    from tests.helpers import load_params
    P = int(z["P"])
    model = HeteGAT_multi().build(P, int(z["F"]), int(z["C"]), (8,), (8, 1), 128, device=dev)
    bp = {k: torch.tensor(z["param_" + k]) for k in ht.PARAM_ORDER}
    load_params(model, bp)
    graphs = [CSRGraph.from_arrays(z[f"rowptr_{p}"], z[f"colidx_{p}"], int(z["N"]), device=dev)
              for p in range(P)]
    x = _t(z["x"], dev)
    with torch.no_grad():
        logits, fe, att = model.inference([x] * P, int(z["C"]), int(z["N"]), False, 0.0, 0.0, graphs,
                                          [8], [8, 1])
    assert np.abs(logits[0].cpu().numpy() - z["logits"]).max() < TOL
    assert np.abs(fe.cpu().numpy() - z["final_embed"]).max() < TOL
    assert np.abs(att.cpu().numpy() - z["att_val"]).max() < TOL


# ------------------------------------------------------------------------ backward
def _oracle_grads(prob, bp, masks=None, keep=1.0, dense=True, table_bf16=False):
    bp = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    xt = torch.tensor(prob["x"][0])
    if dense:
        graphs = [torch.tensor(b[0]) for b in prob["biases"]]
    else:
        graphs = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    logits, _, _ = ht.hetegat_forward([xt] * prob["p"], graphs, bp, keep_in=keep, keep_coef=keep,
                                      masks=masks, dense=dense, table_bf16=table_bf16)
    loss = ht.masked_softmax_cross_entropy(logits, torch.tensor(prob["onehot"]),
                                           torch.tensor(prob["mask"]))
    loss.backward()
    return float(loss.detach()), {k: bp[k].grad.numpy() for k in ht.PARAM_ORDER}, logits.detach().numpy()


def _gpu_loss_and_grads(model, prob, dev, attn_drop=0.0, ffd_drop=0.0):
    from han_amd import layers, ops
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    mask = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    model.zero_grad_flat()
    M = model.node_level([x] * prob["p"], graphs, attn_drop, ffd_drop, True, ops.ACT_ELU)
    Z, _ = model.semantic(M)
    loss, acc, logits = model.classifier_loss(Z, labels, mask, 1.0 / int(prob["mask"].sum()))
    loss.backward()
    grads = {k: getattr(model, k).grad.detach().cpu().numpy().copy() for k in ht.PARAM_ORDER}
    return float(loss), grads, logits.cpu().numpy(), float(acc)


@pytest.mark.parametrize("n,f,p,dens", [(7, 5, 1, [0.4]), (64, 20, 2, [0.1, 0.6]),
                                         (300, 40, 3, [0.005, 0.05, 0.5]),
                                         (150, 24, 8, [0.03, 0.3, 0.1, 0.01])])      # P = 8: configs[4]
def test_gradients_match_autograd_oracle(dev, n, f, p, dens):
    """Hand-written K1/K2/K3/classifier backward vs float64 autograd of the
    dense restatement (SURVEY.md section 8a 'Backward')."""
    prob = make_problem(7 + n, n, f, p, 3, dens)
    model, bp = build_model(prob, dev)
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp)
    loss, grads, lg, acc = _gpu_loss_and_grads(model, prob, dev)
    assert abs(loss - loss_ref) < 1e-4
    assert np.abs(lg - lg_ref).max() < TOL
    acc_ref = float(ht.masked_accuracy(torch.tensor(lg_ref), torch.tensor(prob["onehot"]),
                                       torch.tensor(prob["mask"])))
    assert abs(acc - acc_ref) < 1e-5
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], gref[k]) < GTOL, k


@pytest.mark.parametrize("n,f,p,dens", [(40, 12, 2, [0.1, 0.5]), (200, 30, 1, [0.01]),
                                         (200, 32, 8, [0.03, 0.3, 0.1, 0.01])])     # P = 8: configs[4]
def test_dropout_forward_backward_match_oracle_with_same_masks(dev, n, f, p, dens):
    """Training step with dropout 0.6/0.6: regenerate the kernels' hash masks in
    NumPy, feed them to the oracle, compare loss and every gradient."""
    from han_amd import rng as hrng
    prob = make_problem(33 + n, n, f, p, 3, dens)
    model, bp = build_model(prob, dev)
    drop = 0.6
    hrng.manual_seed(99)
    seeds = [hrng.next_seed() for _ in range(p)]
    hrng.manual_seed(99)                       # the model will draw the same seeds
    keep = rng_ref.keep_prob32(drop)
    masks = []
    for q in range(p):
        rp, ci = ho.bias_to_csr(prob["biases"][q])
        masks.append({"seq": torch.tensor(rng_ref.seq_mask(seeds[q], n, f, 8, drop)),
                      "coef": torch.tensor(rng_ref.coef_mask_csr(seeds[q], rp, ci, 8, drop)),
                      "fts": torch.tensor(rng_ref.fts_mask(seeds[q], n, 64, drop))})
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False)
    loss, grads, lg, _ = _gpu_loss_and_grads(model, prob, dev, attn_drop=drop, ffd_drop=drop)
    assert np.abs(lg - lg_ref).max() < 5 * TOL
    assert abs(loss - loss_ref) < 5e-4
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], gref[k]) < GTOL, k


@pytest.mark.parametrize("in_drop,coef_drop", [(0.0, 0.6), (0.6, 0.0), (0.3, 0.5)])
def test_unequal_dropout_rates_match_oracle(dev, in_drop, coef_drop):
    """ffd_drop != attn_drop (the reference passes them separately, models/gat.py:43-45): the
    generic training instantiations of K2 (only one of the dropouts on, or different keep
    rates) against the oracle fed the same hash masks."""
    from han_amd import rng as hrng
    n, f, p = 80, 10, 2
    prob = make_problem(91, n, f, p, 3, [0.05, 0.4])
    model, bp = build_model(prob, dev)
    hrng.manual_seed(31)
    seeds = [hrng.next_seed() for _ in range(p)]
    hrng.manual_seed(31)
    masks = []
    for q in range(p):
        rp, ci = ho.bias_to_csr(prob["biases"][q])
        mk = {}
        if in_drop > 0:
            mk["seq"] = torch.tensor(rng_ref.seq_mask(seeds[q], n, f, 8, in_drop))
            mk["fts"] = torch.tensor(rng_ref.fts_mask(seeds[q], n, 64, in_drop))
        if coef_drop > 0:
            mk["coef"] = torch.tensor(rng_ref.coef_mask_csr(seeds[q], rp, ci, 8, coef_drop))
        masks.append(mk)
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    graphs = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    keep_in = rng_ref.keep_prob32(in_drop) if in_drop > 0 else 1.0
    keep_coef = rng_ref.keep_prob32(coef_drop) if coef_drop > 0 else 1.0
    logits, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * p, graphs, bpo, keep_in=keep_in,
                                      keep_coef=keep_coef, masks=masks, dense=False)
    loss_ref = ht.masked_softmax_cross_entropy(logits, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    loss, grads, lg, _ = _gpu_loss_and_grads(model, prob, dev, attn_drop=coef_drop, ffd_drop=in_drop)
    assert np.abs(lg - logits.detach().numpy()).max() < 5 * TOL
    assert abs(loss - float(loss_ref)) < 5e-4
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], bpo[k].grad.numpy()) < GTOL, k


def test_dropout_statistics(dev):
    """Keep rate of each stream ~ 0.4 and mean-preserving scaling (1/keep)."""
    from han_amd import ops
    n, f = 4096, 64
    x = torch.ones((n, f), device=dev)
    W = torch.zeros((f, 64), device=dev)
    W[:, ::8] = 1.0                      # column k*8 of head k sums the kept inputs
    z8 = torch.zeros((8, 8), device=dev)
    z = torch.zeros(8, device=dev)
    H, _, _ = ops.project_fwd(x, W, z8, z8, z, z, in_drop=0.6, fts_drop=0.6, seed=12345)
    kept = H[:, ::8] * 0.4               # = number of kept inputs per (row, head)
    rate = float(kept.sum() / (n * f * 8))
    assert abs(rate - 0.4) < 0.003
    ones = int((H.cpu().numpy().view(np.uint32) & 1).sum())
    assert abs(ones / (64.0 * n) - 0.4) < 0.005
    # heads draw different masks
    assert not torch.equal(H[:, 0], H[:, 8])


@pytest.mark.parametrize("K,FP,A", [(16, 4, 128), (4, 16, 128), (2, 32, 128), (1, 64, 128),      # K*F' = 64
                                    (8, 16, 128), (4, 8, 128), (3, 8, 48), (12, 8, 80), (5, 16, 64),  # 128, 32, 24, 96, 80
                                    (1, 4, 16), (2, 64, 128),                                        # 4, 128
                                    # head widths that are not a lane-mapped size run at the next one (12 -> 16,
                                    # 20 -> 32, 3 -> 4, 50 -> 64, 10 -> 16) with zero-weight columns
                                    (5, 12, 128), (3, 20, 64), (2, 3, 32), (1, 50, 100), (10, 10, 128),
                                    # a last layer wider than the K3 / classifier kernels (256, 192 columns)
                                    (8, 32, 128), (3, 64, 40),
                                    (8, 8, 200), (8, 16, 256),                                       # mp_att_size 129 .. 256: K3 kernels (round 3)
                                    (8, 8, 300),                                                     # mp_att_size above the tuned K3 kernels
                                    # round 3: heads wider than 64 columns (hid_units > 64) as slices of one head:
                                    # gathered f2 in K2, shared per-head draws, per-slice projected-row dropout streams
                                    (2, 128, 128), (1, 100, 64), (3, 96, 40), (8, 128, 128)])
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_other_head_shapes_match_oracle(dev, K, FP, A, drop, monkeypatch):
    """hid_units=[F'], n_heads=[K,1] other than 8x8 and mp_att_size other than 128 (models/gat.py:37,42-57
    leave them free).  K*F' = 64: every kernel template (K1 per-head dropout tiles, K2 lane->head maps,
    score-parameter reductions).  Other widths: the heads run through K1/K2 in 64-column groups (short
    groups completed with zero-weight heads), K3 / the classifier take the embedding zero-padded to 64 or
    128 columns (the 128-wide kernels) and mp_att_size zero-padded to 64 / 128 -- exact.  Inference, then
    loss + all gradients with the dropouts on the same hash masks, against the oracle."""
    from han_amd import rng as hrng
    n, f, p = 90, 14, 2
    prob = make_problem(500 + K, n, f, p, 3, [0.06, 0.4], hid_units=[FP], n_heads=(K, 1), mp_att_size=A)
    model, bp = build_model(prob, dev)
    import contextlib

    @contextlib.contextmanager
    def hip_only():      # shapes on the HIP kernels end to end: no library GEMM / torch elementwise in the product path
        with monkeypatch.context() as mp:
            def _no_torch(*a, **k):      # round 3: EVERY shape (any width, any mp_att_size, any head width)
                raise AssertionError("torch.matmul / torch.tanh / softmax in a path the HIP kernels cover")
            for name in ("matmul", "tanh", "softmax", "mm", "bmm", "einsum"):
                mp.setattr(torch, name, _no_torch)
            mp.setattr(torch.nn.functional, "softmax", _no_torch)
            mp.setattr(torch.nn.functional, "cross_entropy", _no_torch)
            yield
    assert (model.K, model.FP, model.A) == (K, FP, A) and tuple(model.w_omega.shape) == (K * FP, A)
    lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * p, 3, n, False, 0.0, 0.0, prob["biases"],
                                             [FP], [K, 1], prob["params"], mp_att_size=A)
    x, graphs = gpu_inputs(prob, dev)
    with torch.no_grad(), hip_only():
        logits, final_embed, att_val = model.inference([x] * p, 3, n, False, 0.0, 0.0, graphs, [FP], [K, 1],
                                                       mp_att_size=A)
    assert tuple(final_embed.shape) == (n, K * FP)
    assert np.abs(logits[0].cpu().numpy() - lg[0]).max() < TOL
    assert np.abs(final_embed.cpu().numpy() - fe).max() < TOL
    assert np.abs(att_val.cpu().numpy() - att).max() < TOL
    hrng.manual_seed(4242)
    seeds = [hrng.next_seed() for _ in range(p)]
    hrng.manual_seed(4242)
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = [group_masks(seeds[q], n, f, K, FP, *ho.bias_to_csr(prob["biases"][q]), drop) for q in range(p)]
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False)
    with hip_only():
        loss, grads, lgg, _ = _gpu_loss_and_grads(model, prob, dev, attn_drop=drop, ffd_drop=drop)
    assert np.abs(lgg - lg_ref).max() < 5 * TOL
    assert abs(loss - loss_ref) < 5e-4
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], gref[k]) < GTOL, k


# ----------------------------------------------------------------------------- K3
@pytest.mark.parametrize("n,p,a,d", [(1, 1, 128, 64), (50, 2, 128, 64), (333, 4, 128, 64), (40, 3, 64, 64),
                                     (2000, 8, 128, 64), (77, 5, 64, 64), (3, 16, 128, 64), (1000, 4, 64, 64),
                                     (5000, 1, 64, 64), (129, 64, 128, 64),
                                     # 128-wide embeddings (hid_units=[16] x 8 heads): the width-templated kernels
                                     (1, 1, 128, 128), (50, 2, 128, 128), (333, 4, 64, 128), (2000, 8, 128, 128),
                                     (77, 5, 128, 128), (129, 64, 64, 128), (5000, 3, 128, 128),
                                     # >= 65536 rows: the forward contraction runs on the bf16 matrix pipe (exact split)
                                     (20000, 4, 128, 64), (70000, 1, 64, 64), (9000, 8, 128, 64), (5000, 16, 64, 64),
                                     (40000, 2, 128, 64),
                                     # round 3: attention spaces of 192 / 256 columns (mp_att_size up to 256) on the
                                     # width-templated kernels, 64- and 128-wide embeddings
                                     (333, 2, 256, 64), (1000, 4, 192, 64), (77, 5, 256, 128), (2000, 8, 192, 128),
                                     (70000, 4, 256, 64), (1, 1, 256, 128),
                                     # round 3: ANY embedding width / attention size (multiples of 64) on the
                                     # run-time-width kernels: 8 x 32 last layers (D = 256), hid_units = [128] x 8 heads
                                     # (D = 1024), mp_att_size = 320 / 512; dW chunks of 128 columns incl. a short one
                                     (300, 3, 128, 256), (257, 4, 320, 192), (100, 2, 512, 1024), (70, 5, 64, 320),
                                     (1, 1, 128, 256), (3000, 4, 256, 256), (129, 64, 64, 192), (50, 2, 320, 64)])
def test_semantic_attention_fwd_bwd(dev, n, p, a, d):
    from han_amd import ops
    rng = np.random.default_rng(n + p)
    M = rng.standard_normal((n, p, d))
    w, b, u = rng.standard_normal((d, a)) * 0.2, rng.standard_normal(a) * 0.2, rng.standard_normal(a)
    Zr, br = ho.simple_att_layer(M, w, b, u, return_alphas=True)
    Z, beta = ops.sem_attn_fwd(_t(M, dev), _t(w, dev), _t(b, dev), _t(u, dev))
    assert np.abs(Z.cpu().numpy() - Zr).max() < TOL
    assert np.abs(beta.cpu().numpy() - br).max() < TOL
    dZ = rng.standard_normal((n, d))
    tM, tw, tb, tu = (torch.tensor(v, requires_grad=True) for v in (M, w, b, u))
    Zt, _ = ht.semantic_attention(tM, tw, tb, tu)
    (Zt * torch.tensor(dZ)).sum().backward()
    dM, dw, db, du = ops.sem_attn_bwd(_t(M, dev), _t(w, dev), _t(b, dev), _t(u, dev), beta, _t(dZ, dev))
    assert rel_err(dM.cpu().numpy(), tM.grad.numpy()) < GTOL
    assert rel_err(dw.cpu().numpy(), tw.grad.numpy()) < GTOL
    assert rel_err(db.cpu().numpy(), tb.grad.numpy()) < GTOL
    assert rel_err(du.cpu().numpy(), tu.grad.numpy()) < GTOL


@pytest.mark.parametrize("n,p,a", [(20000, 4, 128), (9000, 8, 128), (70001, 1, 128), (16400, 4, 64)])
@pytest.mark.parametrize("form", ["g3_f32", "pairs"])
def test_semantic_attention_bwd_measurement_forms(dev, n, p, a, form):
    """The two measurement-only forms of the large-input K3 backward (HAN_FLAG_K3_G3_F32: dW product on the fp32
    pipe, tile by tile; HAN_FLAG_K3_PAIRS: two waves share a tile) against the default form (two tiles per pass,
    dW product on the bf16 pipe) and the fp64 autograd of utils/layers.py:132-164 -- also with an odd number of
    tiles per wave and a last tile of one row (70001 rows)."""
    from han_amd import ops
    rng = np.random.default_rng(n + p)
    M = rng.standard_normal((n, p, 64))
    w, b, u = rng.standard_normal((64, a)) * 0.2, rng.standard_normal(a) * 0.2, rng.standard_normal(a)
    dZ = rng.standard_normal((n, 64))
    gM, gw, gb, gu = _t(M, dev), _t(w, dev), _t(b, dev), _t(u, dev)
    _, beta = ops.sem_attn_fwd(gM, gw, gb, gu)
    flag = ops.FLAG_K3_G3_F32 if form == "g3_f32" else ops.FLAG_K3_PAIRS
    ref = ops.sem_attn_bwd(gM, gw, gb, gu, beta, _t(dZ, dev))
    got = ops.sem_attn_bwd(gM, gw, gb, gu, beta, _t(dZ, dev), flags=flag)
    tM, tw, tb, tu = (torch.tensor(v, requires_grad=True) for v in (M, w, b, u))
    Zt, _ = ht.semantic_attention(tM, tw, tb, tu)
    (Zt * torch.tensor(dZ)).sum().backward()
    for g, r, t64 in zip(got, ref, (tM.grad, tw.grad, tb.grad, tu.grad)):
        assert rel_err(g.cpu().numpy(), t64.numpy()) < GTOL
        assert rel_err(g.cpu().numpy(), r.cpu().numpy()) < GTOL


def test_k2_shared_dropout_hash_is_bitwise_the_default(dev):
    """HAN_FLAG_K2_SHARED_HASH: one attention-dropout hash per lane and 4-edge step (edge q & 3, the lane's head quad),
    the other three edges' words taken from the DPP quad neighbours.  Same draws, same arithmetic: the training forward's
    output and saved state equal the default kernel's bit for bit -- rows of every length up to a few hundred, with the
    degree bins on."""
    from han_amd import ops
    from han_amd.graph import CSRGraph
    g = torch.Generator(device=dev).manual_seed(5)
    n = 3000
    deg = torch.randint(0, 140, (n,), device=dev, generator=g)
    deg[:7] = torch.tensor([0, 1, 3, 4, 63, 64, 65], device=dev)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, n, (int(rowptr[-1]),), device=dev, generator=g).to(torch.int32)
    graph = CSRGraph(rowptr, col, n)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g)
    X, W = rnd(n, 24), rnd(24, 64) * 0.2
    a1, a2, b1, b2, c = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1, rnd(64) * 0.1
    for tdt in (torch.float32, torch.bfloat16):
        H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=11, table_dtype=tdt)
        res = []
        for flag in (False, True):
            ops.K2_SHARED_HASH = flag
            try:
                out, sv = ops.node_attn_fwd(graph, H, f1, a2, b2, c, train=True, coef_drop=0.6, fts_drop=0.6, seed=11)
            finally:
                ops.K2_SHARED_HASH = False
            res.append([out.clone()] + [t.clone() for t in sv if t is not None])
        assert len(res[0]) == len(res[1]) >= 4
        for x, y in zip(*res):
            assert torch.equal(x, y), tdt


def test_empty_inputs_are_noops(dev):
    """N == 0 / E == 0 through the C ABI: every entry point returns without launching
    (edge case of SURVEY.md section 4: empty inputs)."""
    from han_amd import ops
    from han_amd.graph import CSRGraph
    z = lambda *s: torch.zeros(s, device=dev)
    H, f1, f2 = ops.project_fwd(z(0, 7), z(7, 64), z(8, 8), z(8, 8), z(8), z(8))
    assert H.shape == (0, 64) and f1.shape == (0, 8)
    g = CSRGraph(torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev), 0)
    out, _ = ops.node_attn_fwd(g, H, f1, z(8, 8), z(8), z(64))
    assert out.shape == (0, 64)
    Z, beta = ops.sem_attn_fwd(z(0, 2, 64), z(64, 128), z(128), z(128))
    assert Z.shape == (0, 64) and beta.shape == (0, 2)
    # a graph whose rows are all empty: out = act(c) for every row (no neighbour, no softmax)
    g2 = CSRGraph(torch.zeros(6, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev), 5)
    c = torch.linspace(-1, 1, 64, device=dev)
    H5, f15, _ = ops.project_fwd(z(5, 7) + 1, z(7, 64) + 0.1, z(8, 8), z(8, 8), z(8), z(8))
    out2, _ = ops.node_attn_fwd(g2, H5, f15, z(8, 8), z(8), c)
    ref = torch.where(c > 0, c, torch.expm1(c))
    assert torch.allclose(out2, ref[None].expand(5, -1), atol=1e-6)


# ----------------------------------------------------------- classifier / loss / opt
@pytest.mark.parametrize("n,c,hc,d", [(5, 3, 1, 64), (257, 4, 1, 64), (100, 7, 2, 64), (64, 16, 1, 64),
                                      (5, 3, 1, 128), (257, 4, 2, 128), (300, 8, 1, 128), (64, 16, 1, 128),
                                      # more than 16 classes: the class-per-lane kernel
                                      (5, 17, 1, 64), (300, 40, 1, 64), (1000, 64, 2, 64), (257, 64, 2, 128),
                                      (130, 33, 1, 128),
                                      # round 3: any embedding width / class count (head average -> row kernel ->
                                      # Z^T dlogits over the masked rows): 8 x 32 last layers, hid_units = [128],
                                      # widths beyond the register-held row (D = 2048), > 64 classes
                                      (300, 3, 1, 256), (257, 7, 2, 192), (100, 100, 1, 64), (130, 65, 2, 128),
                                      (70, 5, 1, 1024), (40, 3, 1, 2048), (1, 1, 1, 256), (1500, 200, 1, 320)])
def test_classifier_loss_matches_oracle(dev, n, c, hc, d):
    from han_amd import ops
    rng = np.random.default_rng(n + c)
    Z = rng.standard_normal((n, d))
    Wc, bc = rng.standard_normal((hc, d, c)) * 0.3, rng.standard_normal((hc, c)) * 0.1
    labels = rng.integers(0, c, n)
    mask = rng.random(n) < 0.5
    mask[0] = True
    tZ, tW, tb = (torch.tensor(v, requires_grad=True) for v in (Z, Wc, bc))
    logits_t = sum(tZ @ tW[i] + tb[i] for i in range(hc)) / hc
    onehot = torch.tensor(np.eye(c)[labels])
    loss_t = ht.masked_softmax_cross_entropy(logits_t, onehot, torch.tensor(mask))
    acc_t = ht.masked_accuracy(logits_t, onehot, torch.tensor(mask))
    loss_t.backward()
    # also the NumPy restatement (models/base_gattn.py:41-48)
    assert abs(float(loss_t) - ho.masked_softmax_cross_entropy(logits_t.detach().numpy(),
                                                               np.eye(c)[labels], mask)) < 1e-12
    logits, la, grads = ops.classifier_loss(_t(Z, dev), _t(Wc, dev), _t(bc, dev),
                                            _t(labels, dev, torch.int32),
                                            _t(mask.astype(np.uint8), dev, torch.uint8),
                                            1.0 / mask.sum(), backward=True)
    assert np.abs(logits.cpu().numpy() - logits_t.detach().numpy()).max() < TOL
    assert abs(float(la[0]) - float(loss_t)) < 1e-4
    assert abs(float(la[1]) - float(acc_t)) < 1e-5
    dZ, dWc, dbc = grads
    assert rel_err(dZ.cpu().numpy(), tZ.grad.numpy()) < GTOL
    assert rel_err(dWc.cpu().numpy(), tW.grad.numpy()) < GTOL
    assert rel_err(dbc.cpu().numpy(), tb.grad.numpy()) < GTOL
    # the forward-only launch and the backward of the logits alone (han_classifier_bwd) for a caller-supplied dlogits
    logits2, la2, none = ops.classifier_loss(_t(Z, dev), _t(Wc, dev), _t(bc, dev), _t(labels, dev, torch.int32),
                                             _t(mask.astype(np.uint8), dev, torch.uint8), 1.0 / mask.sum())
    assert none is None and torch.equal(logits2, logits) and abs(float(la2[0]) - float(la[0])) < 1e-6
    dl = rng.standard_normal((n, c))
    tZ.grad = tW.grad = tb.grad = None
    logits_u = sum(tZ @ tW[i] + tb[i] for i in range(hc)) / hc
    (logits_u * torch.tensor(dl)).sum().backward()
    dZ, dWc, dbc = ops.classifier_bwd(_t(Z, dev), _t(Wc, dev), _t(bc, dev), _t(dl, dev))
    assert rel_err(dZ.cpu().numpy(), tZ.grad.numpy()) < GTOL
    assert rel_err(dWc.cpu().numpy(), tW.grad.numpy()) < GTOL
    assert rel_err(dbc.cpu().numpy(), tb.grad.numpy()) < GTOL


def test_adam_matches_tf_form(dev):
    from han_amd.base_gattn import TFAdam
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(1000)
    p = _t(p0, dev)
    g = torch.zeros_like(p)
    opt = TFAdam(p, g, lr=0.005, l2_coef=0.001)
    pr, m, v = p0.copy(), np.zeros(1000), np.zeros(1000)
    for t in range(1, 6):
        gr = rng.standard_normal(1000)
        g.copy_(_t(gr, dev))
        opt.step()
        pr, m, v = ho.adam_step_tf(pr, gr + 0.001 * pr, m, v, t)
    assert np.abs(p.cpu().numpy() - pr).max() < 1e-5


# --------------------------------------------------------------------- input format
@pytest.mark.parametrize("n", [1, 5, 64, 129, 1000])
def test_bias_to_csr_gpu(dev, n):
    from han_amd.graph import CSRGraph
    rng = np.random.default_rng(n)
    from tests.helpers import random_adj
    bias = ho.adj_to_bias(random_adj(rng, n, 0.1)[None], [n], 1)
    rp, ci = ho.bias_to_csr(bias)
    g = CSRGraph.from_bias(_t(bias, dev))
    assert np.array_equal(g.rowptr.cpu().numpy(), rp)
    assert np.array_equal(g.colidx.cpu().numpy(), ci)
    gt = g.transpose()
    import scipy.sparse as sp
    a = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n)).T.tocsr()
    a.sort_indices()
    assert np.array_equal(gt.rowptr.cpu().numpy(), a.indptr)
    assert np.array_equal(gt.colidx.cpu().numpy(), a.indices)


# ------------------------------------------------------------------- layer API
def test_attn_head_and_sp_attn_head_api(dev):
    """Reference-named single-head calls (layers.py:7,85) == oracle; attn_head
    on the dense mask == sp_attn_head on the same binary graph."""
    import torch.nn.functional as Fnn
    from han_amd import layers
    prob = make_problem(5, 50, 9, 1, 3, [0.2])
    head = prob["params"]["heads"][0][0]
    ref = ho.attn_head(prob["x"], head, prob["biases"][0])
    params = {k: _t(v, dev) for k, v in head.items()}
    x = _t(prob["x"], dev)
    with torch.no_grad():
        out = layers.attn_head(x, 8, _t(prob["biases"][0], dev), Fnn.elu, params=params)
        rp, ci = ho.bias_to_csr(prob["biases"][0])
        idx = np.stack([np.zeros(len(ci)), np.repeat(np.arange(50), np.diff(rp)), ci])
        sp_adj = torch.sparse_coo_tensor(torch.tensor(idx, dtype=torch.long), torch.ones(len(ci)),
                                         (1, 50, 50)).to(dev)
        out_sp = layers.sp_attn_head(x, 8, sp_adj, Fnn.elu, 50, params=params)
    assert out.shape == (1, 50, 8)
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    assert np.abs(out_sp.cpu().numpy() - ref).max() < TOL
    ref_sp = ho.sp_attn_head(prob["x"], head, rp, ci)
    assert np.abs(out_sp.cpu().numpy() - ref_sp).max() < TOL
    # residual=True with F (9) != out_sz (8): + conv1d(seq, 8, 1) before the ELU (layers.py:38-40)
    rngr = np.random.default_rng(8)
    res = {"W": rngr.standard_normal((9, 8)) * 0.3, "b": rngr.standard_normal(8) * 0.1}
    ref_res = ho.attn_head(prob["x"], head, prob["biases"][0], residual=True, res_params=res)
    with torch.no_grad():
        out_res = layers.attn_head(x, 8, _t(prob["biases"][0], dev), Fnn.elu, residual=True,
                                   params={**params, "res_W": _t(res["W"], dev), "res_b": _t(res["b"], dev)})
    assert np.abs(out_res.cpu().numpy() - ref_res).max() < TOL
    # HAN_nd ablation head (layers.py:49-81): uniform 1/deg weights
    with torch.no_grad():
        out_c = layers.attn_head_const_1(x, 8, _t(prob["biases"][0], dev), Fnn.elu, params=params)
    assert np.abs(out_c.cpu().numpy() - ho.attn_head_const_1(prob["x"], head, prob["biases"][0])).max() < TOL
    # any out_sz up to 64 (layers.py:7 leaves it free): 11 runs at the lane-mapped width 16
    prob11 = make_problem(6, 50, 9, 1, 3, [0.2], hid_units=[11], n_heads=(2, 1))
    head11 = prob11["params"]["heads"][0][1]
    with torch.no_grad():
        out11 = layers.attn_head(_t(prob11["x"], dev), 11, _t(prob11["biases"][0], dev), Fnn.elu,
                                 params={k: _t(v, dev) for k, v in head11.items()})
    assert out11.shape == (1, 50, 11)
    assert np.abs(out11.cpu().numpy() - ho.attn_head(prob11["x"], head11, prob11["biases"][0])).max() < TOL


@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_sp_attn_head_weighted_adjacency_values(dev, drop, monkeypatch):
    """sp_attn_head with NON-binary stored values: they scale the logits,
    e_ij = LeakyReLU(v_ij*f1_i + v_ij*f2_j) (layers.py:95-98) -- forward against
    the NumPy oracle (single head, reference-named call) and all 8 heads + every
    gradient against float64 autograd of the CSR oracle, with and without the
    dropouts (same hash masks), incl. split long rows and the low-degree variant."""
    import torch.nn.functional as Fnn
    from han_amd import layers, ops
    from han_amd.graph import CSRGraph
    monkeypatch.setattr(ops, "SPLIT_DEG", 48)
    monkeypatch.setattr(ops, "SPLIT_CHUNK", 16)
    for n, dens in ((120, 0.3), (90, 0.02)):
        prob = make_problem(77 + n, n, 11, 1, 3, [dens])
        rng = np.random.default_rng(n)
        rp, ci = ho.bias_to_csr(prob["biases"][0])
        vals = rng.uniform(-1.5, 2.0, size=len(ci))           # negative values flip the LeakyReLU branch
        rows = np.repeat(np.arange(n), np.diff(rp))
        if drop == 0.0:
            # reference-named single-head call on a torch sparse (1,N,N) tensor
            head = prob["params"]["heads"][0][0]
            idx = np.stack([np.zeros(len(ci)), rows, ci])
            sp_adj = torch.sparse_coo_tensor(torch.tensor(idx, dtype=torch.long),
                                             torch.tensor(vals, dtype=torch.float32), (1, n, n)).to(dev)
            params = {k: _t(v, dev) for k, v in head.items()}
            with torch.no_grad():
                out = layers.sp_attn_head(_t(prob["x"], dev), 8, sp_adj, Fnn.elu, n, params=params)
            ref = ho.sp_attn_head(prob["x"], head, rp, ci, adj_vals=vals.astype(np.float32).astype(np.float64))
            assert np.abs(out.cpu().numpy() - ref).max() < TOL
        # all K heads through the autograd Function, gradients vs the CSR oracle
        bp = ht.to_batched(prob["params"])
        g = CSRGraph.from_arrays(rp, ci, n, device=dev)
        v32 = torch.tensor(vals, dtype=torch.float32)
        g = CSRGraph(g.rowptr, g.colidx, n, values=v32.to(dev))
        leaf = {k: bp[k][0].clone().to(torch.float32).to(dev).requires_grad_(True)
                for k in ("W", "a1", "b1", "a2", "b2", "c")}
        seed = 0xABCDEF12345
        cfg = {"train": True, "in_drop": drop, "coef_drop": drop, "seeds": (seed,), "act": ops.ACT_ELU,
               "part": None}
        x = _t(prob["x"][0], dev)
        M = layers.NodeLevelAttention.apply(None, *(leaf[k][None] for k in ("W", "a1", "b1", "a2", "b2", "c")),
                                            None, None, (x,), (g,), cfg)
        wgt = torch.tensor(rng.standard_normal((n, 64)), dtype=torch.float32, device=dev)
        (M[:, 0, :] * wgt).sum().backward()
        ref_leaf = {k: bp[k][0].clone().requires_grad_(True) for k in leaf}
        masks, keep = None, 1.0
        if drop > 0:
            keep = rng_ref.keep_prob32(drop)
            masks = {"seq": torch.tensor(rng_ref.seq_mask(seed, n, 11, 8, drop)),
                     "coef": torch.tensor(rng_ref.coef_mask_csr(seed, rp, ci, 8, drop)),
                     "fts": torch.tensor(rng_ref.fts_mask(seed, n, 64, drop))}
        ref_out = ht.node_attention_csr(torch.tensor(prob["x"][0]), torch.tensor(rp), torch.tensor(ci),
                                        ref_leaf["W"], ref_leaf["a1"], ref_leaf["b1"], ref_leaf["a2"],
                                        ref_leaf["b2"], ref_leaf["c"], keep_in=keep, keep_coef=keep,
                                        masks=masks, adj_vals=v32.to(torch.float64))
        (ref_out * wgt.cpu().to(torch.float64)).sum().backward()
        assert np.abs(M[:, 0, :].detach().cpu().numpy() - ref_out.detach().numpy()).max() < 5 * TOL
        for k in leaf:
            assert rel_err(leaf[k].grad.cpu().numpy(), ref_leaf[k].grad.numpy()) < GTOL, k


@pytest.mark.parametrize("K,FP", [(8, 8), (16, 4), (4, 16), (2, 32), (1, 64)])
@pytest.mark.parametrize("drop", [0.0, 0.6])
@pytest.mark.parametrize("dense", [False, True])
def test_lean_kernels_on_small_dense_graphs_match_oracle(dev, K, FP, drop, dense, monkeypatch):
    """HAN_FLAG_LEAN: small graphs with long rows (the reference's own data sets) run K2 on the lean kernels -- scores
    read from the K1 table, softmax in log2 units, one attention-dropout hash per (edge, four heads) handed out by
    ds_bpermute, and for 8 x 8 one lane per head (forward and backward gather).  n = 600, rows of several 64-entry
    pieces with partial last steps; every head shape; inference and loss + all gradients with both dropouts on the
    oracle's exact masks; and the same numbers (to the order of the sums) as the gather kernels."""
    from han_amd import ops, rng as hrng
    if dense and (K, FP) != (8, 8):
        pytest.skip("the matrix-pipe (bit mask) form is built for the reference shape, 8 heads x 8")
    # dense=True: the same graphs through node_attn_dense.h (bit mask + fp32 MFMA tiles, exp-free scores) -- forward,
    # training forward and the transposed-graph backward; dense=False keeps them on the lean CSR kernels
    monkeypatch.setattr(ops, "DENSE", dense)
    monkeypatch.setattr(ops, "DENSE_MIN_DENSITY", 0.1)      # both graphs (80 % and 25 % dense) through the dense form
    n, f, p = 600, 13, 2
    prob = make_problem(300 + K, n, f, p, 3, [0.8, 0.25], hid_units=[FP], n_heads=(K, 1))
    model, bp = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    calls, dcalls = [], []
    real, real_d = ops._use_lean, ops._use_dense
    monkeypatch.setattr(ops, "_use_lean", lambda g, t: calls.append(real(g, t)) or calls[-1])
    monkeypatch.setattr(ops, "_use_dense", lambda g, t, k_, fp_: dcalls.append(real_d(g, t, k_, fp_)) or dcalls[-1])
    lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * p, 3, n, False, 0.0, 0.0, prob["biases"], [FP], [K, 1],
                                             prob["params"])
    with torch.no_grad():
        logits, final_embed, att_val = model.inference([x] * p, 3, n, False, 0.0, 0.0, graphs, [FP], [K, 1])
    assert calls and all(calls)                       # both meta-paths took the lean kernels
    assert dcalls and all(d_ == dense for d_ in dcalls)     # ... or, with dense, the matrix-pipe form
    assert np.abs(logits[0].cpu().numpy() - lg[0]).max() < TOL
    assert np.abs(final_embed.cpu().numpy() - fe).max() < TOL
    monkeypatch.setattr(ops, "LEAN", False)
    with torch.no_grad():      # the gather kernels on the same inputs
        logits_g = model.inference([x] * p, 3, n, False, 0.0, 0.0, graphs, [FP], [K, 1])[0]
    monkeypatch.setattr(ops, "LEAN", True)
    assert not calls[-1] and float((logits_g - logits).abs().max()) < 1e-5
    hrng.manual_seed(777)
    seeds = [hrng.next_seed() for _ in range(p)]
    hrng.manual_seed(777)
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = [group_masks(seeds[q], n, f, K, FP, *ho.bias_to_csr(prob["biases"][q]), drop) for q in range(p)]
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False)
    n_d = len(dcalls)
    loss, grads, lgg, _ = _gpu_loss_and_grads(model, prob, dev, attn_drop=drop, ffd_drop=drop)
    assert calls[-1]
    assert len(dcalls) == n_d + 2 * p and all(d_ == dense for d_ in dcalls[n_d:])      # forward and backward of both meta-paths
    assert np.abs(lgg - lg_ref).max() < 5 * TOL
    assert abs(loss - loss_ref) < 5e-4
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], gref[k]) < GTOL, k


def test_lean_kernels_edge_values_and_ragged_rows(dev):
    """The lean kernels with sp_attn_head's logit-scaling values, rows of very different lengths (empty rows, one
    entry, a full row), unsorted ids and duplicates; forward and backward gather against the gather kernels on the
    same inputs (bitwise-equal masks: the draws are keyed by (row, neighbour, head), not by the order of the walk)."""
    from han_amd import ops
    from han_amd.graph import CSRGraph
    gen = torch.Generator(device=dev).manual_seed(5)
    for n in (256, 257, 700):
        dens = torch.rand((n, 1), device=dev, generator=gen) ** 0.3          # most rows dense, some nearly empty
        dens[3] = 0.0
        dens[n - 1] = 1.0
        adj = torch.rand((n, n), device=dev, generator=gen) < dens
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = adj.sum(1).cumsum(0)
        colidx = adj.nonzero()[:, 1].to(torch.int32).contiguous()
        # reverse the ids inside every row (the lean kernels do not need them sorted) and duplicate one entry
        pos = torch.arange(colidx.numel(), device=dev)
        rows = torch.searchsorted(rowptr, pos, right=True) - 1
        colidx = colidx[rowptr[rows] + rowptr[rows + 1] - 1 - pos].contiguous()
        if colidx.numel() > 10:
            colidx[5] = colidx[4]
        vals = (torch.rand(colidx.numel(), device=dev, generator=gen) * 3 - 1).contiguous()
        for K, FP in ((8, 8), (4, 16)):
            for values in (None, vals):
                g = CSRGraph(rowptr, colidx, n, values=values)
                gt = g.transpose()
                assert g.bitmask() is None and gt.bitmask() is None      # a repeated entry has no bit-mask form: CSR kernels
                a1, a2 = (torch.randn((K, FP), device=dev, generator=gen) * 0.3 for _ in range(2))
                b1, b2 = (torch.randn(K, device=dev, generator=gen) * 0.1 for _ in range(2))
                c = torch.randn(64, device=dev, generator=gen) * 0.1
                X = torch.randn((n, 20), device=dev, generator=gen)
                W = torch.randn((20, 64), device=dev, generator=gen) * 0.2
                dOut = torch.randn((n, 64), device=dev, generator=gen)
                H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.5, fts_drop=0.5, seed=11)
                res = []
                for lean in (True, False):
                    ops.LEAN = lean
                    try:
                        assert ops._use_lean(g, H) == lean and ops._use_lean(gt, H) == lean
                        oe, _ = ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2)
                        ot, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.5, fts_drop=0.5, seed=11, f2=f2)
                        gs, df1, dc = ops.node_attn_bwd_rows(dOut, ot, sv[2], sv[3], f1, sv[1], c, K=K, FP=FP)
                        dH, df2 = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.5, fts_drop=0.5, seed=11)
                    finally:
                        ops.LEAN = True
                    res.append((oe, ot.clone()) + sv[1:] + (dH, df2))
                for a_, b_ in zip(*res):
                    scale = float(b_.abs().max()) + 1.0
                    assert float((a_ - b_).abs().max()) < 5e-5 * scale
                assert float(res[0][0][3].abs().max()) < 1.0 and torch.isfinite(res[0][1]).all()      # the empty row: act(c)


def test_lean_kernels_row_length_edges(dev):
    """Row lengths around every boundary of the lean kernels' walk -- 64-entry pieces, 16-edge steps, the two edges of
    an 8-lane group, the four of a 16-lane group -- for the one-lane-per-head (8 x 8) and the 16-lane (4 x 16) forms,
    forward (eval, training) and backward gather, against the gather kernels."""
    from han_amd import ops
    from han_amd.graph import CSRGraph
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 127, 128, 129, 191,
            192, 193, 255, 256]
    n = 256
    gen = torch.Generator(device=dev).manual_seed(9)
    deg = torch.tensor((lens * ((n + len(lens) - 1) // len(lens)))[:n], device=dev)
    deg = torch.maximum(deg, torch.full_like(deg, 64 + 8))          # keep the mean degree above the lean threshold ...
    deg[:len(lens)] = torch.tensor(lens, device=dev)                 # ... while the first rows take every edge length
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = deg.cumsum(0)
    colidx = torch.cat([torch.randperm(n, generator=gen, device=dev)[:int(d)] for d in deg.tolist()]).to(torch.int32)
    g = CSRGraph(rowptr, colidx.contiguous(), n)
    gt = g.transpose()
    for K, FP in ((8, 8), (4, 16)):
        a1, a2 = (torch.randn((K, FP), device=dev, generator=gen) * 0.3 for _ in range(2))
        b1, b2 = (torch.randn(K, device=dev, generator=gen) * 0.1 for _ in range(2))
        c = torch.randn(64, device=dev, generator=gen) * 0.1
        X = torch.randn((n, 12), device=dev, generator=gen)
        W = torch.randn((12, 64), device=dev, generator=gen) * 0.3
        dOut = torch.randn((n, 64), device=dev, generator=gen)
        H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.3, fts_drop=0.3, seed=21)
        res = {}
        for mode in ("lean", "gather") + (("dense",) if (K, FP) == (8, 8) else ()):
            ops.LEAN, ops.DENSE = mode != "gather", mode == "dense"
            min_density, ops.DENSE_MIN_DENSITY = ops.DENSE_MIN_DENSITY, 0.0      # (the graph is ~35 % dense)
            try:
                assert ops._use_lean(g, H) == ops.LEAN and ops._use_lean(gt, H) == ops.LEAN
                assert ops._use_dense(g, H, K, FP) == ops.DENSE and ops._use_dense(gt, H, K, FP) == ops.DENSE
                oe, _ = ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2)
                oe = oe.clone()
                ot, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.3, fts_drop=0.3, seed=21, f2=f2)
                gs, df1, dc = ops.node_attn_bwd_rows(dOut, ot, sv[2], sv[3], f1, sv[1], c, K=K, FP=FP)
                dH, df2 = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.3, fts_drop=0.3, seed=21)
            finally:
                ops.LEAN = ops.DENSE = True
                ops.DENSE_MIN_DENSITY = min_density
            res[mode] = (oe, ot.clone()) + sv[1:] + (dH, df2)
        for mode in res:      # (the dense form: rows of 0 .. 256 entries in a 256-column table, i.e. empty to full mask rows)
            for name, a_, b_ in zip(("eval", "train", "lse", "aggp", "tsum", "dH", "df2"), res[mode], res["gather"]):
                scale = float(b_.abs().max()) + 1.0
                assert float((a_ - b_).abs().max()) < 5e-5 * scale, (K, FP, mode, name)


def test_dense_form_falls_back_on_the_device_when_the_scores_range_is_wide(dev):
    """The matrix-pipe K2 form shifts every row by a FIXED bound (max f2 of the table), which is exact only while the
    per-head range of f2 stays below 80; beyond that dense_f2_range_kernel leaves a flag in the header and the lean CSR
    kernel behind it -- launched in the same call, predicated on that flag -- produces the result.  Scores scaled so
    that the range is ~400: outputs, saved statistics and gradients equal the gather kernels'; the same inputs at a
    normal scale take the dense kernels (different rounding, same tolerance)."""
    from han_amd import ops, synth
    n = 512
    gen = torch.Generator(device=dev).manual_seed(3)
    g = synth.bernoulli_graph(n, 0.5, 5, dev)
    gt = g.transpose()
    a1, b1, b2 = torch.randn((8, 8), device=dev, generator=gen) * 0.3, torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    c = torch.randn(64, device=dev, generator=gen) * 0.1
    X = torch.randn((n, 16), device=dev, generator=gen)
    W = torch.randn((16, 64), device=dev, generator=gen) * 0.3
    dOut = torch.randn((n, 64), device=dev, generator=gen)
    for scale in (0.3, 60.0):
        a2 = torch.randn((8, 8), device=dev, generator=torch.Generator(device=dev).manual_seed(4)) * scale
        H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.3, fts_drop=0.3, seed=8)
        rng_ = float((f2.max(0).values - f2.min(0).values).max())
        assert (rng_ > 200) == (scale > 1)
        res = {}
        for mode in ("gather", "dense"):
            ops.LEAN, ops.DENSE = mode != "gather", mode == "dense"
            min_density, ops.DENSE_MIN_DENSITY = ops.DENSE_MIN_DENSITY, 0.0
            try:
                assert ops._use_dense(g, H, 8, 8) == ops.DENSE
                oe = ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2)[0].clone()
                ot, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.3, fts_drop=0.3, seed=8, f2=f2)
                gs, df1, _ = ops.node_attn_bwd_rows(dOut, ot, sv[2], sv[3], f1, sv[1], c)
                dH, df2 = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.3, fts_drop=0.3, seed=8)
            finally:
                ops.LEAN = ops.DENSE = True
                ops.DENSE_MIN_DENSITY = min_density
            res[mode] = (oe, ot.clone()) + sv[1:] + (dH, df2)
        for name, a_, b_ in zip(("eval", "train", "lse", "aggp", "tsum", "dH", "df2"), res["dense"], res["gather"]):
            assert torch.isfinite(a_).all(), (scale, name)
            sc = float(b_.abs().max()) + 1.0
            assert float((a_ - b_).abs().max()) < 1e-4 * sc, (scale, name)


def test_dense_form_with_residual_identity_activation_and_stored_pre(dev):
    """The dense K2 form finishes its rows through the same write_row as the CSR kernels: residual term added before the
    activation (layers.py:38-40), identity activation (the kernel then emits the pre-activation for a torch-side callable),
    an output view with a row stride (M[:, p, :]), empty rows, and a table whose last 32-column tile is partial -- against the
    gather kernels on the same inputs, forward and backward."""
    from han_amd import ops, synth
    n = 333                                   # 10 full tiles + one of 13 columns; 5 full 64-row blocks + one of 13 rows
    gen = torch.Generator(device=dev).manual_seed(13)
    g0 = synth.bernoulli_graph(n, 0.7, 9, dev)
    # empty two rows (no self-loop either): their outputs are act(c + res)
    deg = g0.degrees().clone()
    keep = torch.ones(g0.nnz, dtype=torch.bool, device=dev)
    for r in (5, 200):
        keep[int(g0.rowptr[r]):int(g0.rowptr[r + 1])] = False
        deg[r] = 0
    from han_amd.graph import CSRGraph
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = deg.cumsum(0)
    g = CSRGraph(rowptr, g0.colidx[keep].contiguous(), n)
    gt = g.transpose()
    a1, a2 = (torch.randn((8, 8), device=dev, generator=gen) * 0.3 for _ in range(2))
    b1, b2 = (torch.randn(8, device=dev, generator=gen) * 0.1 for _ in range(2))
    c = torch.randn(64, device=dev, generator=gen) * 0.1
    X = torch.randn((n, 24), device=dev, generator=gen)
    W = torch.randn((24, 64), device=dev, generator=gen) * 0.25
    res = torch.randn((n, 64), device=dev, generator=gen) * 0.2
    dOut = torch.randn((n, 64), device=dev, generator=gen)
    H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.4, fts_drop=0.4, seed=17)
    out = {}
    for mode in ("gather", "dense"):
        ops.LEAN, ops.DENSE = mode != "gather", mode == "dense"
        min_density, ops.DENSE_MIN_DENSITY = ops.DENSE_MIN_DENSITY, 0.0
        try:
            assert ops._use_dense(g, H, 8, 8) == ops.DENSE and ops._use_dense(gt, H, 8, 8) == ops.DENSE
            r = []
            for act in (ops.ACT_ELU, ops.ACT_IDENTITY):
                M = torch.zeros((n, 2, 64), device=dev)
                ops.node_attn_fwd(g, H, f1, a2, b2, c, out=M[:, 1, :], activation=act, res=res, f2=f2)
                assert float(M[:, 0, :].abs().max()) == 0.0            # the strided view: nothing written beside it
                Mt = torch.zeros((n, 2, 64), device=dev)
                _, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=Mt[:, 1, :], train=True, coef_drop=0.4, fts_drop=0.4,
                                          seed=17, activation=act, res=res, f2=f2)
                gs, df1, dc = ops.node_attn_bwd_rows(dOut, sv[0], sv[2], sv[3], f1, sv[1], c, activation=act, res=res)
                dH, df2 = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.4, fts_drop=0.4, seed=17)
                r += [M[:, 1, :].clone(), Mt[:, 1, :].clone(), sv[1], sv[2], sv[3], dH, df2, dc]
        finally:
            ops.LEAN = ops.DENSE = True
            ops.DENSE_MIN_DENSITY = min_density
        out[mode] = r
    for i, (a_, b_) in enumerate(zip(out["dense"], out["gather"])):
        assert torch.isfinite(a_).all(), i
        sc = float(b_.abs().max()) + 1.0
        assert float((a_ - b_).abs().max()) < 5e-5 * sc, i
    e = out["dense"][0]                        # ELU eval output of the empty rows: act(c + res)
    for r_ in (5, 200):
        want = torch.nn.functional.elu(c + res[r_])
        assert float((e[r_] - want).abs().max()) < 1e-6


@pytest.mark.parametrize("residual", [False, True])
def test_dense_form_in_a_two_layer_stack(dev, residual, monkeypatch):
    """Two node-attention layers of 8 heads x 8 (models/gat.py:42-57) on 70 % / 90 % dense graphs: every K2 call of both
    layers, forward and backward, takes the dense matrix-pipe form (with the residual term of layers.py:38-40 when asked);
    loss, logits and gradients equal the lean CSR kernels' on the same dropout draws."""
    from han_amd import ops, rng as hrng
    prob = make_problem(5, 400, 11, 2, 3, [0.7, 0.9], hid_units=[8, 8], n_heads=(8, 8, 1), residual=residual)
    model, _ = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    mask = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    res = {}
    for dense in (True, False):
        monkeypatch.setattr(ops, "DENSE", dense)
        calls = []
        real = ops._use_dense
        monkeypatch.setattr(ops, "_use_dense", lambda g, t, k_, fp_, real=real: calls.append(real(g, t, k_, fp_)) or calls[-1])
        hrng.manual_seed(9)
        model.zero_grad_flat()
        M = model.node_level([x] * 2, graphs, 0.5, 0.5, True, ops.ACT_ELU)
        Z, _ = model.semantic(M)
        loss, _, logits = model.classifier_loss(Z, labels, mask, 1.0 / int(prob["mask"].sum()))
        loss.backward()
        monkeypatch.setattr(ops, "_use_dense", real)
        assert len(calls) == 8 and all(c_ == dense for c_ in calls)      # 2 layers x 2 meta-paths x (forward, backward)
        res[dense] = (float(loss.detach()), logits.detach().clone(), model.flat_grad.detach().clone())
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) < 2e-5 and float((a[1] - b[1]).abs().max()) < 1e-4
    assert float((a[2] - b[2]).abs().max()) < 1e-4 * (float(b[2].abs().max()) + 1.0)


def test_return_coef_and_hetegat_class(dev):
    """attn_head(..., return_coef=True) (layers.py:43-44) and HeteGAT.inference(...,
    return_coef=True) (models/gat.py:132-203: shared inputs, head-mean coefficients per
    meta-path) against the dense oracle; with attention dropout the returned
    coefficients are the DROPPED ones (the rebinding at layers.py:30)."""
    import torch.nn.functional as Fnn
    from han_amd import layers
    from han_amd.gat import HeteGAT, HeteGAT_no_coef
    n, p = 60, 2
    prob = make_problem(41, n, 9, p, 3, [0.15, 0.5])
    x = _t(prob["x"], dev)
    head = prob["params"]["heads"][0][0]
    params = {k: _t(v, dev) for k, v in head.items()}
    for drop in (0.0, 0.6):
        seed = 0x5EED5EED
        rp, ci = ho.bias_to_csr(prob["biases"][0])
        masks = None
        if drop > 0:
            cm = np.zeros((1, n, n))
            cm[0, np.repeat(np.arange(n), np.diff(rp)), ci] = rng_ref.coef_mask_csr(seed, rp, ci, 8, drop)[:, 0]
            masks = {"coef": cm}
        ref_out, ref_coef = ho.attn_head(prob["x"], head, prob["biases"][0], coef_drop=drop,
                                         return_coef=True, masks=masks)
        with torch.no_grad():
            out, coefs = layers.attn_head(x, 8, _t(prob["biases"][0], dev), Fnn.elu, coef_drop=drop,
                                          return_coef=True, params=params, training=drop > 0, seed=seed)
        assert coefs.layout == torch.sparse_csr and tuple(coefs.shape) == (n, n)
        if drop > 0:       # float32 keep probability, as the kernels use it
            ref_coef = ref_coef * (1.0 - drop) / rng_ref.keep_prob32(drop)
            ref_out = None
        assert np.abs(coefs.to_dense().cpu().numpy() - ref_coef[0]).max() < 1e-5
        if ref_out is not None:
            assert np.abs(out.cpu().numpy() - ref_out).max() < TOL
    # HeteGAT: one inputs tensor, coef_list[p] = mean over the 8 heads
    HeteGAT.reset_default()
    model = HeteGAT()
    model.build(p, 9, 3, (8,), (8, 1), 128, device=dev)
    load_params(model, ht.to_batched(prob["params"]))
    biases = [_t(b, dev) for b in prob["biases"]]
    with torch.no_grad():
        logits, fe, att, coef_list = model.inference(x, 3, n, False, 0.0, 0.0, biases, [8], [8, 1],
                                                     return_coef=True)
        m2 = HeteGAT_no_coef()
        m2.build(p, 9, 3, (8,), (8, 1), 128, device=dev)
        load_params(m2, ht.to_batched(prob["params"]))
        out3 = m2.inference(x, 3, n, False, 0.0, 0.0, biases, [8], [8, 1])
    lg, fe_ref, att_ref = ho.hetegat_multi_inference([prob["x"]] * p, 3, n, False, 0.0, 0.0, prob["biases"],
                                                     [8], [8, 1], prob["params"])
    assert np.abs(logits.cpu().numpy() - lg).max() < TOL and len(out3) == 3
    assert np.abs(out3[0].cpu().numpy() - lg).max() < TOL
    assert len(coef_list) == p
    for q in range(p):
        per_head = [ho.attn_head(prob["x"], prob["params"]["heads"][q][k], prob["biases"][q],
                                 return_coef=True)[1][0] for k in range(8)]
        assert np.abs(coef_list[q].to_dense().cpu().numpy() - np.mean(per_head, axis=0)).max() < 1e-5


def test_c_abi_demo_program(dev):
    """examples/c_abi_demo.cpp drives K1 -> K2 -> K3 and the K2 / K1 backward chain through
    include/han_hip.h from plain C++ (hipMalloc + a HIP stream; no Python, no torch types at the
    boundary).  Its inputs come from a fixed LCG, rebuilt here; its printed outputs and gradients must
    match the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import __graft_entry__ as entry
    exe = entry.build_c_demo()          # no-op when the binary is newer than its source and the library
    n, f, deg = 300, 24, 5
    r = subprocess.run([exe, str(n), str(f), str(deg)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    vals = np.array([float(t) for t in r.stdout.split()])
    assert vals.size == 2 * n * 64
    M, Z = vals[:n * 64].reshape(n, 64), vals[n * 64:].reshape(n, 64)
    state = np.uint32(12345)

    def lcg(count, scale=1.0):
        nonlocal state
        out = np.empty(count, dtype=np.float32)
        s = int(state)
        for i in range(count):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            out[i] = np.float32(s >> 8) * np.float32(1.0 / 16777216.0) - np.float32(0.5)
        state = np.uint32(s)
        return (out * np.float32(scale)).astype(np.float64)

    X = lcg(n * f).reshape(n, f)
    W = lcg(f * 64, 0.5).reshape(f, 64)
    a1, a2 = lcg(64).reshape(8, 8), lcg(64).reshape(8, 8)
    b1, b2, c = lcg(8, 0.2), lcg(8, 0.2), lcg(64, 0.2)
    wo, bo, uo = lcg(64 * 128, 0.4).reshape(64, 128), lcg(128, 0.2), lcg(128)
    rp = np.arange(0, n * deg + 1, deg)
    ci = ((np.arange(n)[:, None] + np.arange(deg)[None, :]) % n).reshape(-1)
    heads = [{"W": W[:, 8 * k:8 * k + 8], "a1": a1[k], "a2": a2[k], "b1": b1[k], "b2": b2[k],
              "c": c[8 * k:8 * k + 8]} for k in range(8)]
    ref = np.concatenate([ho.sp_attn_head(X[None], h, rp, ci)[0] for h in heads], axis=1)
    assert np.abs(M - ref).max() < TOL
    Zr = ho.simple_att_layer(ref[:, None, :], wo, bo, uo)
    assert np.abs(Z - Zr).max() < TOL

    # the training-side entry points from the same program ("bwd"): K2 forward with the extras ->
    # han_node_attn_bwd_rows -> han_node_attn_bwd_cols (transposed ring) -> han_score_param_bwd ->
    # han_project_bwd, against float64 autograd of the oracle for loss = sum(dOut * out)
    r = subprocess.run([exe, str(n), str(f), str(deg), "bwd"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    vals = np.array([float(t) for t in r.stdout.split()])
    assert vals.size == 2 * n * 64 + f * 64 + 64 + 64 + 8 + 8 + 64
    g = vals[2 * n * 64:]
    got = {"W": g[:f * 64].reshape(f, 64), "a1": g[f * 64:f * 64 + 64].reshape(8, 8),
           "a2": g[f * 64 + 64:f * 64 + 128].reshape(8, 8), "b1": g[f * 64 + 128:f * 64 + 136],
           "b2": g[f * 64 + 136:f * 64 + 144], "c": g[f * 64 + 144:]}
    dOut = lcg(n * 64).reshape(n, 64)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in
          dict(W=W, a1=a1, a2=a2, b1=b1, b2=b2, c=c).items()}
    out = ht.node_attention_csr(torch.tensor(X), torch.tensor(rp), torch.tensor(ci), tp["W"], tp["a1"], tp["b1"],
                                tp["a2"], tp["b2"], tp["c"])
    (out * torch.tensor(dOut)).sum().backward()
    for k, v in got.items():
        assert rel_err(v, tp[k].grad.numpy()) < GTOL, k


def test_errors_are_loud(dev):
    from han_amd import ops
    from han_amd.graph import CSRGraph
    with pytest.raises(ValueError):
        ops.sem_attn_fwd(torch.zeros(4, 2, 64), torch.zeros(64, 128), torch.zeros(128), torch.zeros(128))
    with pytest.raises(ValueError):
        ops.sem_attn_fwd(torch.zeros(4, 2, 64, device=dev, dtype=torch.float64),
                         torch.zeros(64, 128, device=dev), torch.zeros(128, device=dev),
                         torch.zeros(128, device=dev))
    with pytest.raises(NotImplementedError):
        ops._check_heads(4, 8)
    with pytest.raises(ValueError):
        CSRGraph.from_arrays([0, 2], [0, 5], 2, device=dev)     # colidx out of range


# ------------------------------------------------------------------- training loop
def test_three_training_steps_match_oracle(dev):
    """fwd + bwd + L2 + TF-form Adam for 3 steps (dropout 0) == oracle loop
    (ex_acm3025.py:171-218 semantics, models/base_gattn.py:12-24)."""
    from han_amd.trainer import HANTrainer
    prob = make_problem(11, 80, 16, 2, 3, [0.05, 0.4])
    model, bp = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    tm = torch.tensor(prob["mask"].astype(np.uint8))
    vm = torch.tensor((~prob["mask"]).astype(np.uint8))
    tr = HANTrainer(model, [x] * 2, graphs, _t(prob["labels"], dev, torch.int32), tm, vm,
                    lr=0.005, l2_coef=0.001, attn_drop=0.0, ffd_drop=0.0)
    bpo = {k: v.clone() for k, v in bp.items()}
    state = ht.new_adam_state(bpo)
    xt = torch.tensor(prob["x"][0])
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    onehot = torch.tensor(prob["onehot"])
    for _ in range(3):
        tl, ta, vl, va = tr.epoch()
        _, vloss_ref, vacc_ref = ht.train_epoch([xt] * 2, og, bpo, state, onehot,
                                                torch.tensor(prob["mask"]), torch.tensor(~prob["mask"]),
                                                keep=1.0)
        assert abs(float(vl) - vloss_ref) < 2e-4
        assert abs(float(va) - vacc_ref) < 1e-5
    for k in ht.PARAM_ORDER:
        assert np.abs(getattr(model, k).detach().cpu().numpy() - bpo[k].numpy()).max() < 2e-4, k


def test_training_with_dropout_reduces_loss(dev):
    """60 reference epochs (dropout 0.6/0.6, Adam, L2) on a small learnable task: the training loss must
    fall by a large factor and the held-out accuracy rise above chance.  Seeded end to end (variable
    initialisers, dropout seed stream) so that the trajectory does not depend on which tests ran before."""
    from han_amd import rng as hrng, synth
    from han_amd.gat import HeteGAT_multi
    from han_amd.trainer import HANTrainer
    hrng.manual_seed(2)
    wl = synth.make_workload("tiny", device="cpu")
    model = HeteGAT_multi().build(wl["p"], wl["f"], wl["c"], device=dev, generator=torch.Generator().manual_seed(2))
    x = wl["x"].to(dev)
    # make labels learnable: class = argmax of three fixed feature columns
    labels = x[:, :3].argmax(1).to(torch.int32)
    graphs = [g.to(dev) for g in wl["graphs"]]
    tr = HANTrainer(model, [x] * wl["p"], graphs, labels, wl["train_mask"] | 1, wl["val_mask"])
    first = None
    for ep in range(60):
        tl, ta, vl, va = tr.epoch()
        if first is None:
            first = (float(tl), float(vl))
    assert float(tl) < 0.6 * first[0] and np.isfinite(float(vl))
    assert float(vl) < first[1] and float(va) > 0.4          # 3 classes: chance = 1/3


def test_captured_epoch_replay_matches_eager_and_oracle(dev):
    """HANTrainer(use_graph=True): one whole epoch captured into a hipGraph and replayed.
    (a) bitwise the same parameters as the identical device-state flow launched eagerly;
    (b) every replay draws fresh dropout masks and the right Adam bias correction: the
    oracle loop fed masks rebuilt from resolve_seed(fixed seed, device seed word) lands on
    the same parameters."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    n, f, drop = 70, 12, 0.6
    prob = make_problem(52, n, f, 2, 3, [0.08, 0.4])
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    tm = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    vm = _t((~prob["mask"]).astype(np.uint8), dev, torch.uint8)
    trainers = []
    for capture in (True, False):
        model, bp = build_model(prob, dev)
        tr = HANTrainer(model, [x, x], graphs, labels, tm, vm, attn_drop=drop, ffd_drop=drop, use_graph=True)
        tr._capture = capture
        trainers.append(tr)
    bpo = {k: v.clone() for k, v in bp.items()}
    st = ht.new_adam_state(bpo)
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    keep = rng_ref.keep_prob32(drop)
    for ep in range(1, 7):
        outs = []
        for tr in trainers:
            if ep == 1:
                hrng.manual_seed(77)         # both draw the same fixed seeds on their first call
            outs.append([float(v) for v in tr.epoch()])
        assert outs[0] == outs[1], (ep, outs)
        word = int(trainers[0].step_state[0]) & ((1 << 64) - 1)
        assert int(trainers[0].step_state[1]) == ep == trainers[0].opt.t
        fixed = trainers[0].model._fixed_seeds[(0, 2)]
        assert fixed == trainers[1].model._fixed_seeds[(0, 2)]
        masks = []
        for q in range(2):
            sd = rng_ref.resolve_seed(fixed[q], word)
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            masks.append({"seq": torch.tensor(rng_ref.seq_mask(sd, n, f, 8, drop)),
                          "coef": torch.tensor(rng_ref.coef_mask_csr(sd, rp, ci, 8, drop)),
                          "fts": torch.tensor(rng_ref.fts_mask(sd, n, 64, drop))})
        _, vloss, vacc = ht.train_epoch([torch.tensor(prob["x"][0])] * 2, og, bpo, st,
                                        torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]),
                                        torch.tensor(~prob["mask"]), keep=keep, masks=masks)
        assert abs(outs[0][2] - vloss) < 5e-4, ep
    assert trainers[0]._graph is not None and trainers[1]._graph is None
    assert torch.equal(trainers[0].model.flat, trainers[1].model.flat)
    for k in ht.PARAM_ORDER:
        assert np.abs(getattr(trainers[0].model, k).detach().cpu().numpy() - bpo[k].numpy()).max() < 2e-4, k


def test_eval_forward_leaves_the_gradient_buffer_alone(dev):
    """The classifier of an eval forward (torch.no_grad()) neither runs the gradient half of its kernel nor -- in the
    trainer's direct-gradient mode -- writes Wc.grad / bc.grad: ctx.needs_input_grad reports requires_grad whatever
    the grad mode, so layers.classifier_loss_any hands the grad mode to the Function.  (Until round 4 every eval
    forward rewrote those two slices of the flat gradient buffer: harmless after Adam, a race beside a training step.)"""
    from han_amd import layers
    from han_amd.trainer import HANTrainer
    prob = make_problem(61, 80, 12, 2, 3, [0.1, 0.4])
    x, graphs = gpu_inputs(prob, dev)
    model, _ = build_model(prob, dev)
    tr = HANTrainer(model, [x, x], graphs, _t(prob["labels"], dev, torch.int32),
                    _t(prob["mask"].astype(np.uint8), dev, torch.uint8), attn_drop=0.0, ffd_drop=0.0)
    tr.train_step()
    marker = torch.full_like(model.flat_grad, 7.0)
    model.flat_grad.copy_(marker)
    vl, va = tr.eval_step()
    assert torch.equal(model.flat_grad, marker) and np.isfinite(float(vl))
    Z = torch.randn(80, 64, device=dev)
    with torch.no_grad():
        l0 = layers.classifier_loss_any(Z, model.Wc, model.bc, tr.labels, tr.train_mask, tr.w_train)[0]
    assert torch.equal(model.flat_grad, marker)
    l1 = layers.classifier_loss_any(Z, model.Wc, model.bc, tr.labels, tr.train_mask, tr.w_train)[0]
    assert float(l0) == float(l1) and not torch.equal(model.flat_grad, marker)     # with grad mode on it does write


@pytest.mark.parametrize("form", ["branch", "sections"])
@pytest.mark.parametrize("shape", ["small", "acm"])
def test_overlap_eval_branch_of_the_captured_epoch(dev, shape, form):
    """HANTrainer(use_graph=True, overlap_eval="branch" | "sections"): the eval forward of the parameters an epoch
    starts with runs beside the training step inside the captured graph -- as one branch that only Adam waits for, or
    in two pieces inside the training step's own fork / join sections.  Against the plain captured epoch: bit-equal training pairs and parameters in every replay (the eval branch has scratch buffers of its own and
    never reads a half-updated parameter), the validation pair of epoch k - 1 in call k (the branch projects all
    meta-paths in one fused K1 launch instead of one launch per path stream, hence a tolerance on that pair)."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    if shape == "small":
        prob = make_problem(58, 90, 14, 3, 3, [0.05, 0.3, 0.6])
    else:                       # the ACM-like shape class: split-F K1 (F = 1870) and the lean K2 kernels (24 % dense)
        prob = make_problem(59, 3025, 1870, 2, 3, [0.003, 0.24])   # at size: the branches really run side by side
    P = len(prob["biases"])
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    tm = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    vm = _t((~prob["mask"]).astype(np.uint8), dev, torch.uint8)
    hist, trainers = [], []
    for overlap in (False, True):
        model, _ = build_model(prob, dev)
        hrng.manual_seed(99)
        tr = HANTrainer(model, [x] * P, graphs, labels, tm, vm, attn_drop=0.6, ffd_drop=0.6, use_graph=True,
                        overlap_eval=form if overlap else False)
        hist.append([[float(v) for v in tr.epoch()] for _ in range(7)])
        for _ in range(120):             # replays issued back to back, no host sync in between (as bench.py does)
            tr.epoch()
        torch.cuda.synchronize()
        trainers.append(tr)
    plain, over = hist
    assert trainers[1]._graph is not None and trainers[1]._eval_stream is not None
    for k in range(7):
        assert over[k][:2] == plain[k][:2], (k, over[k], plain[k])
        if k:
            assert abs(over[k][2] - plain[k - 1][2]) < 1e-5 * max(1.0, abs(plain[k - 1][2])), k
            assert abs(over[k][3] - plain[k - 1][3]) < 1e-6, k
    assert torch.equal(trainers[0].model.flat, trainers[1].model.flat)
    vl, va = trainers[1].flush_eval()
    vl0, va0 = trainers[0].eval_step()
    assert abs(float(vl) - float(vl0)) < 1e-5 * max(1.0, abs(float(vl0))) and abs(float(va) - float(va0)) < 1e-6
    assert torch.equal(trainers[1]._flat_prev, trainers[1].model.flat)


def test_path_streams_do_not_change_the_numbers(dev, monkeypatch):
    """a captured epoch forks one stream per meta-path (HAN_PATH_STREAMS, default on).  The same epochs on
    the single chain (HAN_PATH_STREAMS=0) must give bit-equal parameters and losses: the streams only reorder
    independent launches; forward and backward of a meta-path use the stream recorded at forward time."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    prob = make_problem(53, 90, 14, 3, 3, [0.05, 0.3, 0.6])
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    tm = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    flats, hist = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("HAN_PATH_STREAMS", flag)
        model, _ = build_model(prob, dev)
        tr = HANTrainer(model, [x] * 3, graphs, labels, tm, attn_drop=0.6, ffd_drop=0.6, use_graph=True)
        assert (model.path_streams is not None) == (flag == "1")
        hrng.manual_seed(31)
        hist.append([[float(v) for v in tr.epoch()] for _ in range(5)])
        torch.cuda.synchronize()
        assert tr._graph is not None
        flats.append(model.flat.detach().clone())
    assert hist[0] == hist[1]
    assert torch.equal(flats[0], flats[1])


@pytest.mark.parametrize("residual", [False, True])
def test_side_stream_does_not_change_the_numbers(dev, residual):
    """HANTrainer(side_stream=True): the backward's dW of meta-path p runs on a second stream beside the gather of
    meta-path p + 1 (layers._on_side).  Same epochs on one stream: bit-equal losses and parameters -- the stream only
    moves independent launches; dH is handed to it with record_stream, its scratch buffers are its own."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    prob = make_problem(54, 300, 40, 3, 3, [0.05, 0.3, 0.6], residual=residual)
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    tm = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    flats, hist = [], []
    for flag in (True, False):
        model, _ = build_model(prob, dev)
        tr = HANTrainer(model, [x] * 3, graphs, labels, tm, attn_drop=0.6, ffd_drop=0.6, side_stream=flag)
        assert (model.side_stream is not None) == flag
        hrng.manual_seed(32)
        hist.append([[float(v) for v in tr.epoch()] for _ in range(6)])
        torch.cuda.synchronize()
        flats.append(model.flat.detach().clone())
    assert hist[0] == hist[1]
    assert torch.equal(flats[0], flats[1])
    # off by default; "auto" stays off at this size (on from 262 144 rows)
    for kw in ({}, {"side_stream": "auto"}):
        model, _ = build_model(prob, dev)
        HANTrainer(model, [x] * 3, graphs, labels, tm, **kw)
        assert model.side_stream is None


def test_workspace_growth_between_graph_replays(dev):
    """The process-global scratch buffers are grow-only and a captured epoch replays with the raw
    pointers it saw: a LARGER request for the same buffers between two replays (another model, a
    bigger graph) must not free memory the graph still uses.  Capture a small epoch, then run a much
    larger model eagerly (every workspace tag grows), then replay: bitwise the eager flow."""
    from han_amd import ops, rng as hrng, synth
    from han_amd.gat import HeteGAT_multi
    from han_amd.trainer import HANTrainer
    n, f, drop = 70, 12, 0.6
    prob = make_problem(52, n, f, 2, 3, [0.08, 0.4])
    x, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    tm = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    trainers = []
    for capture in (True, False):
        model, _ = build_model(prob, dev)
        tr = HANTrainer(model, [x, x], graphs, labels, tm, attn_drop=drop, ffd_drop=drop, use_graph=True)
        tr._capture = capture
        trainers.append(tr)

    def both():
        outs = [[float(v) for v in tr.epoch()] for tr in trainers]
        assert outs[0] == outs[1]
    hrng.manual_seed(3)
    trainers[0].epoch()
    hrng.manual_seed(3)
    trainers[1].epoch()
    both(); both()                     # trainers[0] has captured and replayed once
    assert trainers[0]._graph is not None
    before = {k: v.data_ptr() for k, v in ops._workspaces.items()}
    # every scratch buffer the capture saw is outgrown (a larger request for the same tag replaces it) ...
    for (d, tag), t in list(ops._workspaces.items()):
        ops._ws(2 * t.numel() + 1, torch.device(d), tag)
    assert all(before[k] != v.data_ptr() for k, v in ops._workspaces.items() if k in before) and before
    # ... and a much larger model then works in the new buffers
    wl = synth.make_workload("syn-100k", device=dev, n_override=300_000)
    big = HeteGAT_multi().build(wl["p"], wl["f"], wl["c"], device=dev)
    tb = HANTrainer(big, [wl["x"]] * wl["p"], wl["graphs"], wl["labels"], wl["train_mask"], wl["val_mask"])
    tb.epoch()
    torch.cuda.synchronize()
    del tb, big, wl
    torch.cuda.empty_cache()
    junk = torch.full((64 << 20,), 7.0e30, device=dev)      # anything freed would be handed out again here
    both(); both()
    assert torch.equal(trainers[0].model.flat, trainers[1].model.flat)
    del junk


def test_inference_with_arbitrary_activation_callables(dev):
    """models/gat.py:36: `activation` may be any callable (the reference's GAT class passes
    lambda x: x).  Anything but ELU / identity is applied by torch on the kernels' pre-activation, per
    head; forward against the oracle and the gradients through torch's autograd of the callable."""
    from han_amd import layers, ops
    prob = make_problem(78, 60, 10, 2, 3, [0.1, 0.4])
    model, bp = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    sm_np = lambda a: np.exp(a - a.max(-1, keepdims=True)) / np.exp(a - a.max(-1, keepdims=True)).sum(-1, keepdims=True)
    for tact, nact in ((torch.tanh, np.tanh), (lambda t: torch.softmax(t, -1), sm_np), (lambda t: t, lambda a: a)):
        lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * 2, 3, 60, False, 0.0, 0.0, prob["biases"], [8], [8, 1],
                                                 prob["params"], activation=nact)
        with torch.no_grad():
            logits, final_embed, att_val = model.inference([x[None]] * 2, 3, 60, False, 0.0, 0.0, graphs, [8], [8, 1],
                                                           activation=tact)
        assert np.abs(logits.cpu().numpy() - lg).max() < TOL
        assert np.abs(final_embed.cpu().numpy() - fe).max() < TOL
        assert np.abs(att_val.cpu().numpy() - att).max() < TOL
    # gradients with tanh: float64 autograd of the torch restatement with the same activation swapped in
    import torch.nn.functional as Fnn
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    orig = Fnn.elu
    try:
        ht.Fnn.elu = torch.tanh
        lref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * 2, og, bpo)
    finally:
        ht.Fnn.elu = orig
    loss_ref = ht.masked_softmax_cross_entropy(lref, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    model.zero_grad_flat()
    code, post = layers._act_code(torch.tanh)
    M = model.node_level([x, x], graphs, 0.0, 0.0, True, code, post=post)
    Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
    loss, _, _ = layers.ClassifierLoss.apply(Z, model.Wc, model.bc, _t(prob["labels"], dev, torch.int32),
                                             _t(prob["mask"].astype(np.uint8), dev, torch.uint8),
                                             1.0 / int(prob["mask"].sum()))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-4
    for k in ht.PARAM_ORDER:
        assert rel_err(getattr(model, k).grad.cpu().numpy(), bpo[k].grad.numpy()) < GTOL, k


def test_training_learns_planted_communities(dev):
    """End-to-end sanity beyond parity: on a synthetic task WITH structure (communities visible
    in the meta-path graphs, features = noisy community indicators) the reference training
    recipe (lr 0.005, l2 0.001, dropout 0.6/0.6, early stopping on the val split) must reach a
    test accuracy far above chance, and the semantic attention must favour the cleaner meta-path."""
    from han_amd import rng as hrng, synth
    from han_amd.gat import HeteGAT_multi
    from han_amd.trainer import HANTrainer
    torch.manual_seed(0)
    hrng.manual_seed(5)
    wl = synth.planted_partition(1500, 4, 2, 32, deg_in=8, deg_out=2, noise=1.0, seed=3, device=dev)
    model = HeteGAT_multi().build(2, 32, 4, (8,), (8, 1), 128, device=dev)
    tr = HANTrainer(model, [wl["x"]] * 2, wl["graphs"], wl["labels"], wl["train_mask"], wl["val_mask"],
                    patience=40, use_graph=True)
    for epoch in range(200):
        tl, ta, vl, va = (float(v) for v in tr.epoch())
        if tr.early_stopping(vl, va):
            break
    tr.restore_best()
    w = 1.0 / int(wl["test_mask"].sum())
    _, test_acc = tr.eval_step(wl["test_mask"], w)
    assert float(test_acc) > 0.9, float(test_acc)          # chance is 0.25; measured 0.977
    with torch.no_grad():
        _, _, att = model.inference([wl["x"]] * 2, 4, 1500, False, 0.0, 0.0, wl["graphs"], [8], [8, 1])
    m = att.mean(0)
    assert float(m[0]) > float(m[1])          # meta-path 0 has fewer cross-community edges


# ------------------------------------------------------------------- skewed graphs
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_row_split_for_long_rows_matches_oracle(dev, monkeypatch, drop):
    """Rows longer than SPLIT_DEG go through the chunk + finish kernels (forward
    and backward gather): same loss and gradients as the oracle, and as the
    unsplit kernels."""
    from han_amd import ops, rng as hrng
    prob = make_problem(77, 300, 20, 2, 3, [0.02, 0.5])     # meta-path 1: ~150 neighbours, hub: 300
    model, bp = build_model(prob, dev)
    hrng.manual_seed(123)
    base_loss, base_grads, base_lg, _ = _gpu_loss_and_grads(model, prob, dev, drop, drop)
    monkeypatch.setattr(ops, "SPLIT_DEG", 16)
    monkeypatch.setattr(ops, "SPLIT_CHUNK", 24)
    hrng.manual_seed(123)
    loss, grads, lg, _ = _gpu_loss_and_grads(model, prob, dev, drop, drop)
    x, graphs = gpu_inputs(prob, dev)
    sp = graphs[1].row_split(16, 24)
    assert sp is not None and sp["n_long"] > 250 and sp["n_chunks"] > 2 * sp["n_long"]
    assert np.abs(lg - base_lg).max() < 2e-5 and abs(loss - base_loss) < 2e-5
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], base_grads[k]) < 2e-4, k
    if drop == 0.0:
        loss_ref, gref, lg_ref = _oracle_grads(prob, bp)
        assert np.abs(lg - lg_ref).max() < TOL
        for k in ht.PARAM_ORDER:
            assert rel_err(grads[k], gref[k]) < GTOL, k


def _mixed_degree_graph(n, transpose, seed=5):
    """Rows of 1, 3, 15, 16, 17, 64 and 500 distinct neighbours mixed at random + three rows of 20 000 (beyond
    SPLIT_DEG): every bin of the degree-binned launch (short: 16-lane group per row; mid: wave per row; long:
    chunks) and the boundaries 15 / 16 / 17 between the first two.  Distinct ids per row by construction (an
    arithmetic progression with a stride coprime to n), ascending.  transpose: the same edges with the roles
    swapped, so that the BACKWARD's graph (the transposed one) has this degree mix."""
    rng = np.random.default_rng(seed)
    assert n % 2 == 0 and n % 5 == 0 and n > 20000
    degs = rng.choice([1, 3, 15, 16, 17, 64, 500], size=n, p=[0.2, 0.2, 0.15, 0.15, 0.15, 0.1, 0.05])
    degs[[7, n // 3, n - 2]] = 20000
    rows, cols = [], []
    for i, d in enumerate(degs):
        step = int(rng.choice([1, 3, 7, 9, 11, 13]))                # odd, not a multiple of 5: coprime to n = 2^a 5 b...
        start = int(rng.integers(0, n))
        cc = (start + step * np.arange(d, dtype=np.int64)) % n
        rows.append(np.full(d, i, dtype=np.int64))
        cols.append(np.sort(cc))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    if transpose:
        order = np.lexsort((rows, cols))
        rows, cols = cols[order], rows[order]
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rp[1:])
    return rp, cols.astype(np.int32), degs


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_degree_binned_launch_on_mixed_rows(dev, drop, transpose, monkeypatch):
    """K2 launches are degree-binned per (graph, direction).  A graph mixing rows of 1, 3, 15, 16,
    17, 64, 500 and 20 000 entries: loss, logits and every gradient against the oracle (fed the kernels' own dropout
    masks), against the un-binned launch, and bitwise determinism of the binned one (forward outputs and gradients)."""
    import math
    from han_amd import ops, rng as hrng
    from han_amd.graph import CSRGraph
    from han_amd.gat import HeteGAT_multi
    n, f, c = 20480, 12, 3
    assert math.gcd(n, 3 * 7 * 9 * 11 * 13) == 1
    rp, ci, degs = _mixed_degree_graph(n, transpose)
    rng = np.random.default_rng(11)
    prob = dict(n=n, f=f, p=1, c=c, x=0.5 * rng.standard_normal((1, n, f)),
                params=ho.init_params(rng, 1, f, c, nonzero_biases=True))
    labels = rng.integers(0, c, size=n)
    mask = rng.random(n) < 0.3
    model, bp = build_model(prob, dev)
    g = CSRGraph.from_arrays(rp, ci, n, device=dev)
    for gg in (g, g.transpose()):                      # both directions carry bins; the mixed one has all three
        rb, sp = gg.row_bins(ops.SHORT_DEG, ops.SPLIT_DEG), gg.row_split(ops.SPLIT_DEG, ops.SPLIT_CHUNK)
        assert rb["n_short"] + rb["n_mid"] + (sp["n_long"] if sp else 0) == n
    mixed = g.transpose() if transpose else g
    rb, sp = mixed.row_bins(ops.SHORT_DEG, ops.SPLIT_DEG), mixed.row_split(ops.SPLIT_DEG, ops.SPLIT_CHUNK)
    assert sp["n_long"] == 3 and rb["n_short"] == int((degs < 16).sum()) and rb["n_mid"] == int((degs >= 16).sum()) - 3
    short = rb["short_rows"].cpu().numpy()
    assert (np.diff((degs[short] + 3) // 4) >= 0).all()            # ordered by the number of 4-entry steps
    x = _t(prob["x"][0], dev)
    lab_t, mask_t = _t(labels, dev, torch.int32), _t(mask.astype(np.uint8), dev, torch.uint8)

    def run():
        hrng.manual_seed(4242)
        model.zero_grad_flat()
        M = model.node_level([x], [g], drop, drop, True, ops.ACT_ELU)
        Z, _ = model.semantic(M)
        loss, _, logits = model.classifier_loss(Z, lab_t, mask_t, 1.0 / int(mask.sum()))
        loss.backward()
        return (float(loss), logits.detach().cpu().numpy(), M.detach().clone(),
                {k: getattr(model, k).grad.detach().clone() for k in ht.PARAM_ORDER})

    loss, lg, M1, g1 = run()
    loss2, lg2, M2, g2 = run()
    assert torch.equal(M1, M2) and all(torch.equal(g1[k], g2[k]) for k in ht.PARAM_ORDER)      # bitwise reproducible
    monkeypatch.setattr(ops, "BINNED", False)
    loss_u, lg_u, M_u, g_u = run()
    monkeypatch.setattr(ops, "BINNED", True)
    assert np.abs(lg - lg_u).max() < 2e-5 and abs(loss - loss_u) < 2e-5
    for k in ht.PARAM_ORDER:
        assert rel_err(g1[k].cpu().numpy(), g_u[k].cpu().numpy()) < 2e-4, k
    # the oracle, fed the masks the kernels drew
    masks, keep = None, 1.0
    if drop > 0:
        hrng.manual_seed(4242)
        seed = hrng.next_seed()
        keep = rng_ref.keep_prob32(drop)
        masks = [{"seq": torch.tensor(rng_ref.seq_mask(seed, n, f, 8, drop)),
                  "coef": torch.tensor(rng_ref.coef_mask_csr(seed, rp, ci, 8, drop)),
                  "fts": torch.tensor(rng_ref.fts_mask(seed, n, 64, drop))}]
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    logits_o, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])], [(torch.tensor(rp), torch.tensor(ci))], bpo,
                                        keep_in=keep, keep_coef=keep, masks=masks, dense=False)
    loss_o = ht.masked_softmax_cross_entropy(logits_o, torch.tensor(np.eye(c)[labels]), torch.tensor(mask))
    loss_o.backward()
    assert np.abs(lg - logits_o.detach().numpy()).max() < (5 * TOL if drop else TOL)
    assert abs(loss - float(loss_o)) < 5e-4
    for k in ht.PARAM_ORDER:
        assert rel_err(g1[k].cpu().numpy(), bpo[k].grad.numpy()) < GTOL, k


# ------------------------------------------------- BASELINE.json configs at full size
ACM_NNZ = (29281, 2210761)                  # entries incl. self-loops of PAP / PSP
DBLP_NNZ = (11113, 5000495, 12924399)       # APA / APCPA / APTPA
def test_acm_like_config_logits_parity(dev):
    """configs[1]: ACM3025 shape (N=3025, P=2, F=1870, 8 heads x 8, C=3), DENSE
    bias_mat path, fp32, logits within 1e-4 of the float64 oracle (synthetic data
    with the measured PAP/PSP densities: ACM3025.mat is not available offline)."""
    from han_amd import synth
    n, f, p = 3025, 1870, 2
    prob = make_problem(3025, n, f, p, 3, None, nnz=ACM_NNZ)        # PAP / PSP entry counts, exactly
    assert [int((b > -1e8).sum()) for b in prob["biases"]] == list(ACM_NNZ)
    prob["x"] *= 0.2                        # bag-of-words-like magnitudes keep |logits| O(1)
    lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * p, 3, n, False, 0.0, 0.0, prob["biases"],
                                             [8], [8, 1], prob["params"])
    model, _ = build_model(prob, dev)
    x = _t(prob["x"], dev)
    biases = [_t(b, dev) for b in prob["biases"]]          # (1,N,N) fp32 additive masks, as fed to TF
    with torch.no_grad():
        logits, final_embed, att_val = model.inference([x] * 3, 3, n, False, 0.0, 0.0, biases, [8], [8, 1])
    assert np.abs(logits.cpu().numpy() - lg).max() < TOL
    assert np.abs(final_embed.cpu().numpy() - fe).max() < TOL
    assert np.abs(att_val.cpu().numpy() - att).max() < TOL


def test_dblp_like_config_sparse_path_parity(dev):
    """configs[2]: DBLP4057 shape (N=4057, P=3, F=334, C=4), sparse CSR path with the
    measured APA / APCPA / APTPA degree mix (mean 2.7 / 1233 / 3186)."""
    from han_amd import layers
    n, f, p = 4057, 334, 3
    prob = make_problem(4057, n, f, p, 4, None, nnz=DBLP_NNZ)       # APA / APCPA / APTPA entry counts, exactly
    assert [int((b > -1e8).sum()) for b in prob["biases"]] == list(DBLP_NNZ)
    prob["x"] *= 0.3
    bp = ht.to_batched(prob["params"])
    graphs_o = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    with torch.no_grad():
        lg, fe, att = ht.hetegat_forward([torch.tensor(prob["x"][0])] * p, graphs_o, bp)
    model, _ = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    assert graphs[0].nnz < 12 * n < graphs[1].nnz            # both the low-degree and the wave-per-row kernels
    with torch.no_grad():
        logits, final_embed, att_val = model.inference([x] * p, 4, n, False, 0.0, 0.0, graphs, [8], [8, 1])
    assert np.abs(logits[0].cpu().numpy() - lg.numpy()).max() < TOL
    assert np.abs(final_embed.cpu().numpy() - fe.numpy()).max() < TOL
    assert np.abs(att_val.cpu().numpy() - att.numpy()).max() < TOL


def test_syn1m_size_independent_properties(dev):
    """configs[3] at FULL size (N = 1M, deg 50): properties that need no oracle.
    (1) softmax rows sum to 1: with H == 1 every output is exactly act(1 + c);
    (2) bitwise determinism of two launches (no float atomics);
    (3) with a2 = 0 the scores do not depend on H, so the pre-activation is linear
        in H: K2(2 H1 + 3 H2) == 2 K2(H1) + 3 K2(H2) - 4 c;
    (4) backward gather pass: sum_j dH_j = sum_i g_i when every alpha-weight path is
        switched off except the direct term (a1 = a2 = 0, uniform attention)."""
    from han_amd import ops, synth
    n = 1_000_000
    g = synth.random_regular_graph(n, 50, 99, dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    f1 = torch.randn((n, 8), device=dev, generator=gen)
    a2 = torch.randn((8, 8), device=dev, generator=gen)
    b2 = torch.randn(8, device=dev, generator=gen)
    c = torch.randn(64, device=dev, generator=gen) * 0.1
    ones = torch.ones((n, 64), device=dev)
    out, _ = ops.node_attn_fwd(g, ones, f1, a2, b2, c, activation=ops.ACT_IDENTITY)
    assert float((out - (1.0 + c)[None]).abs().max()) < 2e-6
    H1 = torch.randn((n, 64), device=dev, generator=gen)
    H2 = torch.randn((n, 64), device=dev, generator=gen)
    o1, _ = ops.node_attn_fwd(g, H1, f1, a2, b2, c)
    o1b, _ = ops.node_attn_fwd(g, H1, f1, a2, b2, c)
    assert torch.equal(o1, o1b)
    z = torch.zeros_like(a2)
    l1, _ = ops.node_attn_fwd(g, H1, f1, z, b2, c, activation=ops.ACT_IDENTITY)
    l2, _ = ops.node_attn_fwd(g, H2, f1, z, b2, c, activation=ops.ACT_IDENTITY)
    l12, _ = ops.node_attn_fwd(g, 2 * H1 + 3 * H2, f1, z, b2, c, activation=ops.ACT_IDENTITY)
    assert float((l12 - (2 * l1 + 3 * l2 - 4 * c[None])).abs().max()) < 1e-4
    # backward: uniform attention (f1 = 0, a = 0) -> alpha_ij = 1/50; dH_j = sum_i g_i / 50
    zero8 = torch.zeros((n, 8), device=dev)
    _, saved = ops.node_attn_fwd(g, H1, zero8, z, torch.zeros(8, device=dev), c, train=True)
    pre, lse, aggp, tsum = saved
    dOut = torch.randn((n, 64), device=dev, generator=gen)
    gs, df1, dc = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, zero8, lse, c)
    gg, stats = ops.gs_views(gs)
    dH, df2 = ops.node_attn_bwd_cols(g.transpose(), gs, H1, zero8, df1, z, z)
    assert abs(float(dH.double().sum()) - float(gg.double().sum())) < 1e-3 * float(gg.double().abs().sum()) ** 0.5 + 1.0
    assert float((dc.double() - gg.double().sum(0)).abs().max()) < 5e-2


# ------------------------------------------------------------------------ bf16 tables
def test_node_attn_bf16_table_exact_against_rounded_inputs(dev):
    """bf16 storage, fp32 accumulate: fed the bf16-rounded rows, the oracle must be
    matched to fp32 accuracy (the only error is the storage rounding of H itself)."""
    from han_amd import ops
    rng = np.random.default_rng(12)
    n = 400
    bias, rp, ci, H, f1, a2, b2, c, g = _k2_inputs(rng, n, 0.1, dev)
    Hb = _t(H, dev).to(torch.bfloat16)
    Hr = Hb.to(torch.float32).cpu().numpy().astype(np.float64)          # what the kernel reads
    ref, pre_ref, lse_ref = _k2_oracle(bias, Hr, f1, _f2(Hr, a2, b2), c)
    out, saved = ops.node_attn_fwd(g, Hb, _t(f1, dev), _t(a2, dev), _t(b2, dev), _t(c, dev), train=True)
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    assert saved[0].data_ptr() == out.data_ptr()           # the output itself (the pre-activation is not stored)
    o64 = out.cpu().numpy().astype(np.float64)
    assert np.abs(np.where(o64 > 0, o64, np.log1p(np.minimum(o64, 0))) - pre_ref).max() < 10 * TOL


@pytest.mark.parametrize("P", [2, 8])
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_bf16_mode_forward_backward(dev, drop, P):
    """configs[4] storage: X and the H / g tables in bf16.  Forward within bf16
    rounding of the fp64 oracle; gradients within 3e-2 relative (bf16 g table).
    P = 8 is configs[4]'s meta-path count: it drives K3's P = 8 wave-local kernels
    (forward and backward) through the model, with regenerated dropout masks."""
    from han_amd import rng as hrng
    from han_amd.gat import HeteGAT_multi
    from tests.helpers import load_params
    prob = make_problem(91, 200, 32, P, 3, [0.03, 0.3, 0.1, 0.01])
    xb = torch.tensor(prob["x"][0], dtype=torch.float32).to(torch.bfloat16)
    prob["x"] = xb.to(torch.float32).numpy().astype(np.float64)[None]     # the oracle sees the same bf16 features
    bp = ht.to_batched(prob["params"])
    model = HeteGAT_multi().build(P, 32, 3, device=dev, table_dtype=torch.bfloat16)
    load_params(model, bp)
    masks, keep = None, 1.0
    hrng.manual_seed(31)
    if drop > 0:
        seeds = [hrng.next_seed() for _ in range(P)]
        hrng.manual_seed(31)
        keep = rng_ref.keep_prob32(drop)
        masks = []
        for q in range(P):
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            masks.append({"seq": torch.tensor(rng_ref.seq_mask(seeds[q], 200, 32, 8, drop)),
                          "coef": torch.tensor(rng_ref.coef_mask_csr(seeds[q], rp, ci, 8, drop)),
                          "fts": torch.tensor(rng_ref.fts_mask(seeds[q], 200, 64, drop))})
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False)
    from han_amd import layers, ops
    _, graphs = gpu_inputs(prob, dev)
    labels = _t(prob["labels"], dev, torch.int32)
    mask = _t(prob["mask"].astype(np.uint8), dev, torch.uint8)
    model.zero_grad_flat()
    xg = xb.to(dev)
    M = model.node_level([xg] * P, graphs, drop, drop, True, ops.ACT_ELU)
    Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
    loss, acc, logits = layers.ClassifierLoss.apply(Z, model.Wc, model.bc, labels, mask,
                                                    1.0 / int(prob["mask"].sum()))
    loss.backward()
    # (a) against the oracle that restates the bf16 STORAGE (H rows rounded to bf16 with the keep bit in
    # the lowest mantissa bit, g rounded to bf16) and computes everything else in float64: pins the
    # kernels' arithmetic in this mode; what is left is fp32 accumulation + a handful of elements whose
    # fp32-vs-fp64 value straddles a bf16 rounding boundary
    loss_q, gq, lg_q = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False, table_bf16=True)
    assert rel_err(logits.cpu().numpy(), lg_q) < 2e-3
    assert abs(float(loss) - loss_q) < 2e-3 * max(1.0, abs(loss_q))
    for k in ht.PARAM_ORDER:
        assert rel_err(getattr(model, k).grad.cpu().numpy(), gq[k]) < 1e-2, k
    # (b) against the plain float64 oracle: the price of bf16 storage itself (~0.4 % per stored element,
    # ~1.2 % in training where the lowest mantissa bit carries the keep bit -- an effective 6-bit
    # mantissa).  Informational bound; (a) is the parity statement.  The score gradients (a1, b1, b2)
    # are sums of softmax-gradient terms that nearly cancel, so storage noise shows up larger there
    # (measured: up to 10 % of the largest element at P = 8 with dropout, 1-3 % elsewhere).
    assert rel_err(logits.cpu().numpy(), lg_ref) < 4e-2
    assert abs(float(loss) - loss_ref) < 4e-2 * max(1.0, abs(loss_ref))
    for k in ht.PARAM_ORDER:
        got = getattr(model, k).grad.cpu().numpy()
        assert rel_err(got, gref[k]) < (0.2 if k in ("a1", "b1", "b2") else 6e-2), k


@pytest.mark.parametrize("deg", [5, 40, 300])
@pytest.mark.parametrize("tdt", [torch.float32, torch.bfloat16])
def test_masked_edges_backward_is_bit_identical(dev, deg, tdt):
    """HAN_FLAG_MASKED_EDGES: destinations whose g row is identically zero (outside the loss mask of a one-layer
    model) are replaced by -1 IN PLACE in the transposed graph; the kernel loads nothing for them and adds the
    remaining terms in the positions and order of the full pass.  dH and df2 must be BIT-EQUAL to the full
    graph run through the same (general, non-FAST: table_gid given) instantiation, low- and high-degree rows,
    weighted adjacency, fp32 and bf16 tables; and equal the FAST full pass to rounding."""
    from han_amd import ops, synth
    from han_amd.graph import CSRGraph
    n = 900
    g = synth.random_regular_graph(n, deg, 11, dev)
    rng = np.random.default_rng(deg)
    gens = lambda *sh: _t(rng.standard_normal(sh), dev)
    a1, a2, b1, b2, c = gens(8, 8) * 0.3, gens(8, 8) * 0.3, gens(8) * 0.1, gens(8) * 0.1, gens(64) * 0.1
    H, f1, f2 = ops.project_fwd(gens(n, 64), torch.eye(64, device=dev), a1, a2, b1, b2, fts_drop=0.6, seed=9,
                                table_dtype=tdt)
    live = _t((rng.random(n) < 0.15).astype(np.float32), dev) > 0
    dOut = gens(n, 64) * live[:, None]                                   # g == 0 outside the mask
    for vals in (None, _t(rng.uniform(0.5, 1.5, g.nnz), dev)):
        gg = CSRGraph(g.rowptr, g.colidx, n, validate=False, values=vals)
        gt = gg.transpose()
        _, sv = ops.node_attn_fwd(gg, H, f1, a2, b2, c, train=True, coef_drop=0.6, fts_drop=0.6, seed=77)
        pre, lse, aggp, tsum = sv
        gs, df1, _ = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, f1, lse, c, table_dtype=tdt)
        ident = torch.arange(n, dtype=torch.int32, device=dev)             # table_gid = identity: the general instantiation
        kw = dict(coef_drop=0.6, fts_drop=0.6, seed=77)
        dH_f, df2_f = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, table_gid=ident, **kw)
        gm = gt.with_masked_columns(live)
        assert int((gm.colidx < 0).sum()) > 0.5 * gm.nnz
        dH_m, df2_m = ops.node_attn_bwd_cols(gm, gs, H, f2, df1, a1, a2, table_gid=ident, **kw)
        assert torch.equal(dH_f, dH_m) and torch.equal(df2_f, df2_m), vals is not None
        dH_q, df2_q = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, **kw)   # FAST full pass
        assert float((dH_q - dH_m).abs().max()) < 1e-5 * max(1.0, float(dH_q.abs().max()))
        assert float((df2_q - df2_m).abs().max()) < 1e-5 * max(1.0, float(df2_q.abs().max()))


@pytest.mark.parametrize("K,FP", [(4, 16), (16, 4), (2, 32), (1, 64), (5, 12), (12, 8)])
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_bf16_mode_other_head_shapes(dev, K, FP, drop):
    """bf16 tables for head shapes other than 8 x 8 (every lane map of K2 and K1's bf16 epilogue, head
    groups, padded head widths), against the oracle that restates the bf16 storage."""
    from han_amd import ops, rng as hrng
    from han_amd.gat import HeteGAT_multi
    from tests.helpers import load_params
    n, f, P = 150, 20, 2
    prob = make_problem(400 + K, n, f, P, 3, [0.05, 0.3], hid_units=[FP], n_heads=(K, 1))
    xb = torch.tensor(prob["x"][0], dtype=torch.float32).to(torch.bfloat16)
    prob["x"] = xb.to(torch.float32).numpy().astype(np.float64)[None]
    bp = ht.to_batched(prob["params"])
    model = HeteGAT_multi().build(P, f, 3, (FP,), (K, 1), device=dev, table_dtype=torch.bfloat16)
    load_params(model, bp)
    masks, keep = None, 1.0
    hrng.manual_seed(77)
    if drop > 0:
        seeds = [hrng.next_seed() for _ in range(P)]
        hrng.manual_seed(77)
        keep = rng_ref.keep_prob32(drop)
        masks = [group_masks(seeds[q], n, f, K, FP, *ho.bias_to_csr(prob["biases"][q]), drop) for q in range(P)]
    loss_q, gq, lg_q = _oracle_grads(prob, bp, masks=masks, keep=keep, dense=False, table_bf16=True)
    _, graphs = gpu_inputs(prob, dev)
    model.zero_grad_flat()
    M = model.node_level([xb.to(dev)] * P, graphs, drop, drop, True, ops.ACT_ELU)
    Z, _ = model.semantic(M)
    loss, acc, logits = model.classifier_loss(Z, _t(prob["labels"], dev, torch.int32),
                                              _t(prob["mask"].astype(np.uint8), dev, torch.uint8),
                                              1.0 / int(prob["mask"].sum()))
    loss.backward()
    assert rel_err(logits.cpu().numpy(), lg_q) < 2e-3
    assert abs(float(loss) - loss_q) < 2e-3 * max(1.0, abs(loss_q))
    for k in ht.PARAM_ORDER:
        assert rel_err(getattr(model, k).grad.cpu().numpy(), gq[k]) < 1e-2, k


def _bf16_to_f64(t):
    return t.to(torch.float32).cpu().numpy().astype(np.float64)


def _k2_sampled_rows_reference(rows, rowptr, colidx, Hb, f1, a2, b2, c, seed, coef_drop, fts_drop, slope=0.2):
    """Oracle arithmetic of utils/layers.py:26-35 for a few destination rows of a huge graph,
    fed the masks the kernels draw: the attention-dropout draws regenerated from (seed, i, j)
    (tests/rng_ref.py) and the projected-row keep bits read from the stored rows (bit 0 of
    the bf16 mantissa).  Hb: the bf16 table on the GPU.  Returns pre-activation rows and lse."""
    keep_c = rng_ref.keep_prob32(coef_drop) if coef_drop > 0 else 1.0
    keep_f = rng_ref.keep_prob32(fts_drop) if fts_drop > 0 else 1.0
    a2 = np.asarray(a2, np.float64); b2 = np.asarray(b2, np.float64); c = np.asarray(c, np.float64)
    K, FP = a2.shape
    pre = np.zeros((len(rows), K * FP)); lse = np.zeros((len(rows), K))
    rp = rowptr.cpu().numpy()
    for n_, i in enumerate(rows):
        js = colidx[int(rp[i]):int(rp[i + 1])].long()
        Hj_t = Hb[js]
        Hj = _bf16_to_f64(Hj_t).reshape(len(js), K, FP)
        keepbits = (Hj_t.view(torch.int16).cpu().numpy().astype(np.int64) & 1).reshape(len(js), K, FP)
        f2 = (Hj * a2[None]).sum(-1) + b2[None]                       # scores from the UNDROPPED stored rows
        x = np.asarray(f1[i].cpu().numpy(), np.float64)[None] + f2    # (deg, K)
        e = np.where(x > 0, x, slope * x)
        m = e.max(0)
        pexp = np.exp(e - m)
        lse[n_] = m + np.log(pexp.sum(0))
        alpha = pexp / pexp.sum(0)
        if coef_drop > 0:
            alpha = alpha * rng_ref.coef_draws(seed, np.full(len(js), i), js.cpu().numpy(), K, coef_drop) / keep_c
        Hd = Hj * keepbits / keep_f if fts_drop > 0 else Hj
        pre[n_] = (alpha[:, :, None] * Hd).sum(0).reshape(-1) + c
    return pre, lse


def test_syn10m_bf16_table_full_size(dev):
    """configs[4] at FULL size on one table: N = 10M rows, deg 50, bf16 rows (the 1.28 GB table is
    HBM-served, U = 8 eval unroll, FAST training instantiation).  Size-independent properties
    (row sums, determinism, linearity, sum_j dH_j = sum_i g_i) plus sampled rows against the
    oracle arithmetic with the regenerated dropout masks, forward and backward."""
    from han_amd import ops, synth
    n, deg = 10_000_000, 50
    g = synth.random_regular_graph(n, deg, 4242, dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    f1 = torch.randn((n, 8), device=dev, generator=gen)
    a2 = torch.randn((8, 8), device=dev, generator=gen) * 0.5
    b2 = torch.randn(8, device=dev, generator=gen)
    c = torch.randn(64, device=dev, generator=gen) * 0.1
    z8 = torch.zeros_like(a2)
    bf = torch.bfloat16
    # (1) rows of the softmax sum to 1: H == 1 -> out == 1 + c
    ones = torch.ones((n, 64), device=dev, dtype=bf)
    out, _ = ops.node_attn_fwd(g, ones, f1, a2, b2, c, activation=ops.ACT_IDENTITY)
    assert float((out - (1.0 + c)[None]).abs().max()) < 2e-6
    del ones, out
    # (2) determinism, (3) linearity with a2 = 0 on integer-valued tables (exact in bf16)
    H1 = torch.randint(-8, 9, (n, 64), device=dev, generator=gen).to(bf)
    H2 = torch.randint(-8, 9, (n, 64), device=dev, generator=gen).to(bf)
    o1, _ = ops.node_attn_fwd(g, H1, f1, a2, b2, c)
    o1b, _ = ops.node_attn_fwd(g, H1, f1, a2, b2, c)
    assert torch.equal(o1, o1b)
    del o1, o1b
    l1, _ = ops.node_attn_fwd(g, H1, f1, z8, b2, c, activation=ops.ACT_IDENTITY)
    l2, _ = ops.node_attn_fwd(g, H2, f1, z8, b2, c, activation=ops.ACT_IDENTITY)
    H12 = (2 * H1.float() + 3 * H2.float()).to(bf)
    l12, _ = ops.node_attn_fwd(g, H12, f1, z8, b2, c, activation=ops.ACT_IDENTITY)
    assert float((l12 - (2 * l1 + 3 * l2 - 4 * c[None])).abs().max()) < 1e-3
    del l1, l2, l12, H12, H2
    # (4) eval-mode rows against the oracle arithmetic (no dropout)
    rs = np.random.default_rng(3)
    rows = sorted(int(r) for r in rs.integers(0, n, 24)) + [0, n - 1]
    Hr = (torch.randn((n, 64), device=dev, generator=gen)).to(bf)
    oe, _ = ops.node_attn_fwd(g, Hr, f1, a2, b2, c, activation=ops.ACT_IDENTITY)
    pre_ref, _ = _k2_sampled_rows_reference(rows, g.rowptr, g.colidx, Hr, f1, a2.cpu().numpy(), b2.cpu().numpy(),
                                            c.cpu().numpy(), 0, 0.0, 0.0)
    assert np.abs(oe[rows].cpu().numpy() - pre_ref).max() < TOL
    del oe
    # (5) training launch, both dropouts on (FAST instantiation): the stored LSBs are the keep bits
    seed, drop = 0x1234_5678_9ABC, 0.6
    ot, saved = ops.node_attn_fwd(g, Hr, f1, a2, b2, c, train=True, coef_drop=drop, fts_drop=drop, seed=seed,
                                  activation=ops.ACT_IDENTITY)
    ot2, saved2 = ops.node_attn_fwd(g, Hr, f1, a2, b2, c, train=True, coef_drop=drop, fts_drop=drop, seed=seed,
                                    activation=ops.ACT_IDENTITY)
    assert torch.equal(ot, ot2) and all(torch.equal(x, y) for x, y in zip(saved, saved2))
    del ot2, saved2
    pre_ref, lse_ref = _k2_sampled_rows_reference(rows, g.rowptr, g.colidx, Hr, f1, a2.cpu().numpy(),
                                                  b2.cpu().numpy(), c.cpu().numpy(), seed, drop, drop)
    assert np.abs(ot[rows].cpu().numpy() - pre_ref).max() < 5 * TOL
    pre, lse, aggp, tsum = saved
    assert np.abs(pre[rows].cpu().numpy() - pre_ref).max() < 5 * TOL
    assert np.abs(lse[rows].cpu().numpy() - lse_ref).max() < TOL
    o3, _ = ops.node_attn_fwd(g, Hr, f1, a2, b2, c, train=True, coef_drop=drop, fts_drop=drop, seed=seed + 1,
                              activation=ops.ACT_IDENTITY)
    assert not torch.equal(ot, o3)
    del o3, ot
    # (6) backward at full size: row-local half, then the transposed-graph gather
    a1 = torch.randn((8, 8), device=dev, generator=gen) * 0.5
    dOut = torch.randint(-4, 5, (n, 64), device=dev, generator=gen).float()     # exact in the bf16 g table
    gs, df1, dc = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, f1, lse, c, activation=ops.ACT_IDENTITY,
                                         table_dtype=bf)
    gg, stats = ops.gs_views(gs, 8, 8, bf)
    assert torch.equal(gg.float(), dOut)
    gt = g.transpose()
    f2 = (Hr.float().view(n, 8, 8) * a2[None]).sum(-1) + b2[None]
    dH, df2 = ops.node_attn_bwd_cols(gt, gs, Hr, f2, df1, a1, a2, coef_drop=drop, fts_drop=drop, seed=seed)
    dHb, df2b = ops.node_attn_bwd_cols(gt, gs, Hr, f2, df1, a1, a2, coef_drop=drop, fts_drop=drop, seed=seed)
    assert torch.equal(dH, dHb) and torch.equal(df2, df2b)
    del dHb, df2b
    # sampled source rows j against the formulas of SURVEY.md 8a "Backward":
    #   df2_j = sum_i alpha_ij s'_ij (m_ij/keep * g_i . H~_j - s_i),  dH_j = keep_j/keep * sum_i a~_ij g_i + df1_j a1 + df2_j a2
    keep = rng_ref.keep_prob32(drop)
    a1n, a2n = a1.cpu().numpy().astype(np.float64), a2.cpu().numpy().astype(np.float64)
    cp = gt.rowptr.cpu().numpy()
    for j in rows[::3]:
        ii = gt.colidx[int(cp[j]):int(cp[j + 1])].long()
        if ii.numel() == 0:
            continue
        st = stats[ii].cpu().numpy().astype(np.float64)                 # (deg, K, 4): f1_i, lse_i, s_i
        gi = _bf16_to_f64(gg[ii]).reshape(-1, 8, 8)
        Hj = _bf16_to_f64(Hr[j]).reshape(8, 8)
        kb = (Hr[j].view(torch.int16).cpu().numpy().astype(np.int64) & 1).reshape(8, 8)
        Hd = Hj * kb / keep
        x = st[:, :, 0] + f2[j].cpu().numpy().astype(np.float64)[None]
        sg = np.where(x > 0, 1.0, 0.2)
        alpha = np.exp(np.where(x > 0, x, 0.2 * x) - st[:, :, 1])
        am = rng_ref.coef_draws(seed, ii.cpu().numpy(), np.full(ii.numel(), j), 8, drop) / keep
        dot = (gi * Hd[None]).sum(-1)
        df2_ref = (alpha * sg * (am * dot - st[:, :, 2])).sum(0)
        acc = ((alpha * am)[:, :, None] * gi).sum(0)
        dH_ref = acc * kb / keep + df1[j].cpu().numpy().astype(np.float64)[:, None] * a1n + df2_ref[:, None] * a2n
        assert np.abs(df2[j].cpu().numpy() - df2_ref).max() < 2e-3 * max(1.0, np.abs(df2_ref).max())
        assert np.abs(dH[j].cpu().numpy().reshape(8, 8) - dH_ref).max() < 2e-3 * max(1.0, np.abs(dH_ref).max())
    del dH, df2, stats, gg, gs
    # (7) sum_j dH_j = sum_i g_i under uniform attention, no dropout (a = 0, f1 = 0)
    zero8 = torch.zeros((n, 8), device=dev)
    _, saved = ops.node_attn_fwd(g, H1, zero8, z8, torch.zeros(8, device=dev), c, train=True)
    pre, lse, aggp, tsum = saved
    gs, df1, dc = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, zero8, lse, c, table_dtype=bf)
    gg, _ = ops.gs_views(gs, 8, 8, bf)
    dH, _ = ops.node_attn_bwd_cols(gt, gs, H1, zero8, df1, z8, z8)
    tot = float(gg.double().abs().sum())
    assert abs(float(dH.double().sum()) - float(gg.double().sum())) < 1e-6 * tot + 1.0
    assert float((dc.double() - gg.double().sum(0)).abs().max()) < 1e-6 * tot / 64 + 1.0


# ------------------------------------------------------------------------ multi-layer
@pytest.mark.parametrize("drop,residual", [(0.0, False), (0.6, False), (0.0, True), (0.6, True)])
def test_multi_layer_stack_matches_oracle(dev, drop, residual):
    """models/gat.py:48-57 with hid_units=[8,16], n_heads=[8,4,1]: forward through
    inference(), and loss + every gradient (incl. han_project_bwd_input, which carries
    the second layer's gradient back into the first) against the float64 oracle."""
    from han_amd import layers, ops, rng as hrng
    prob = make_problem(62, 150, 11, 2, 3, [0.05, 0.4], hid_units=[8, 16], n_heads=(8, 4, 1),
                        residual=residual)
    model, bp = build_model(prob, dev)
    assert hasattr(model, "Wr_1") == residual
    x, graphs = gpu_inputs(prob, dev)
    if drop == 0:
        lg_np, fe_np, att_np = ho.hetegat_multi_inference([prob["x"]] * 2, 3, 150, False, 0.0, 0.0,
                                                          prob["biases"], [8, 16], [8, 4, 1], prob["params"],
                                                          residual=residual)
        with torch.no_grad():
            logits, fe, att = model.inference([x[None]] * 2, 3, 150, False, 0.0, 0.0, graphs, [8, 16], [8, 4, 1],
                                              residual=residual)
        assert np.abs(logits.cpu().numpy() - lg_np).max() < TOL
        assert np.abs(fe.cpu().numpy() - fe_np).max() < TOL
    hrng.manual_seed(19)
    seeds = [hrng.next_seed() for _ in range(4)]
    hrng.manual_seed(19)
    model.zero_grad_flat()
    M = model.node_level([x, x], graphs, drop, drop, True, ops.ACT_ELU)
    Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
    loss, _, logits = layers.ClassifierLoss.apply(Z, model.Wc, model.bc, _t(prob["labels"], dev, torch.int32),
                                                  _t(prob["mask"].astype(np.uint8), dev, torch.uint8),
                                                  1.0 / int(prob["mask"].sum()))
    loss.backward()
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = []
        for q in range(2):
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            mk = lambda sd, f, K: {"seq": torch.tensor(rng_ref.seq_mask(sd, 150, f, K, drop)),
                                   "coef": torch.tensor(rng_ref.coef_mask_csr(sd, rp, ci, K, drop)),
                                   "fts": torch.tensor(rng_ref.fts_mask(sd, 150, 64, drop))}
            m0 = mk(seeds[q], 11, 8)
            m0["layers"] = [mk(seeds[2 + q], 64, 4)]
            masks.append(m0)
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    lg_ref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * 2, og, bpo, keep_in=keep,
                                      keep_coef=keep, masks=masks)
    loss_ref = ht.masked_softmax_cross_entropy(lg_ref, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    assert rel_err(logits.cpu().numpy(), lg_ref.detach().numpy()) < 1e-4
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 5e-4 * max(1.0, float(loss_ref.detach()))
    for k in ht.param_order(bp):
        assert rel_err(getattr(model, k).grad.cpu().numpy(), bpo[k].grad.numpy()) < GTOL, k


@pytest.mark.parametrize("drop,residual", [(0.0, False), (0.6, False), (0.6, True)])
def test_wide_head_in_a_deeper_layer(dev, drop, residual, monkeypatch):
    """hid_units=[8,96], n_heads=[8,2,1] (models/gat.py:48-57 leaves the widths free): the second layer's heads
    are 96 columns wide -- two 64-column slices of one head each (layers.WideHeadAttention: shared scores, gathered
    f2, shared per-head draws) -- and send their input gradient back into the first layer; the last layer is 192
    columns wide (run-time-width K3 / classifier kernels).  Loss and every gradient against the float64 oracle on
    the same hash masks, with no torch GEMM / softmax anywhere in the product path."""
    from han_amd import layers, ops, rng as hrng
    n = 120
    prob = make_problem(63, n, 11, 2, 3, [0.05, 0.4], hid_units=[8, 96], n_heads=(8, 2, 1), residual=residual)
    model, bp = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    hrng.manual_seed(23)
    seeds = [hrng.next_seed() for _ in range(4)]
    hrng.manual_seed(23)
    model.zero_grad_flat()
    with monkeypatch.context() as mp:
        def _no_torch(*a, **k):
            raise AssertionError("torch GEMM / softmax in the product path")
        for name in ("matmul", "tanh", "softmax", "mm", "bmm", "einsum"):
            mp.setattr(torch, name, _no_torch)
        M = model.node_level([x, x], graphs, drop, drop, True, ops.ACT_ELU)
        assert tuple(M.shape) == (n, 2, 192)
        Z, _ = model.semantic(M)
        loss, _, logits = model.classifier_loss(Z, _t(prob["labels"], dev, torch.int32),
                                                _t(prob["mask"].astype(np.uint8), dev, torch.uint8),
                                                1.0 / int(prob["mask"].sum()))
        loss.backward()
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = []
        for q in range(2):
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            m0 = group_masks(seeds[q], n, 11, 8, 8, rp, ci, drop)
            m0["layers"] = [group_masks(seeds[2 + q], n, 64, 2, 96, rp, ci, drop)]
            masks.append(m0)
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    lg_ref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * 2, og, bpo, keep_in=keep, keep_coef=keep, masks=masks)
    loss_ref = ht.masked_softmax_cross_entropy(lg_ref, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    assert rel_err(logits.cpu().numpy(), lg_ref.detach().numpy()) < 1e-4
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 5e-4 * max(1.0, float(loss_ref.detach()))
    for k in ht.param_order(bp):
        assert rel_err(getattr(model, k).grad.cpu().numpy(), bpo[k].grad.numpy()) < GTOL, k


def test_wide_single_head_api_and_bf16_tables(dev):
    """attn_head(seq, out_sz = 100, ...) (utils/layers.py:7 leaves out_sz free) with return_coef and the residual
    branch, and a model with 128-wide heads on bf16 tables (same slices, bf16 rows)."""
    import torch.nn.functional as Fnn
    from han_amd import layers
    n = 50
    prob = make_problem(7, n, 9, 1, 3, [0.2], hid_units=[100], n_heads=(1, 1))
    head = prob["params"]["heads"][0][0]
    params = {k: _t(v, dev) for k, v in head.items()}
    x = _t(prob["x"], dev)
    ref, ref_coef = ho.attn_head(prob["x"], head, prob["biases"][0], return_coef=True)
    with torch.no_grad():
        out, coefs = layers.attn_head(x, 100, _t(prob["biases"][0], dev), Fnn.elu, params=params, return_coef=True)
    assert out.shape == (1, n, 100)
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    assert np.abs(coefs.to_dense().cpu().numpy() - ref_coef[0]).max() < 1e-5
    rngr = np.random.default_rng(8)
    res = {"W": rngr.standard_normal((9, 100)) * 0.3, "b": rngr.standard_normal(100) * 0.1}
    ref_res = ho.attn_head(prob["x"], head, prob["biases"][0], residual=True, res_params=res)
    with torch.no_grad():
        out_res = layers.attn_head(x, 100, _t(prob["biases"][0], dev), Fnn.elu, residual=True,
                                   params={**params, "res_W": _t(res["W"], dev), "res_b": _t(res["b"], dev)})
    assert np.abs(out_res.cpu().numpy() - ref_res).max() < TOL
    # bf16 tables: forward within the bf16 storage error of the fp32-table model, gradients finite and close
    from han_amd.gat import HeteGAT_multi
    prob2 = make_problem(8, 80, 12, 2, 3, [0.1, 0.3], hid_units=[128], n_heads=(2, 1))
    m32, bp = build_model(prob2, dev)
    m16 = HeteGAT_multi().build(2, 12, 3, (128,), (2, 1), 128, device=dev, table_dtype=torch.bfloat16)
    load_params(m16, bp)
    x2, graphs2 = gpu_inputs(prob2, dev)
    with torch.no_grad():
        l32 = m32.inference([x2] * 2, 3, 80, False, 0.0, 0.0, graphs2, [128], [2, 1])[0]
        l16 = m16.inference([x2] * 2, 3, 80, False, 0.0, 0.0, graphs2, [128], [2, 1])[0]
    assert rel_err(l16.cpu().numpy(), l32.cpu().numpy()) < 2e-2


def test_locality_pass_is_a_pure_relabelling(dev):
    """han_amd.reorder on the real kernels: a breadth-first relabelling of a graph whose locality hides
    behind shuffled ids changes nothing but the order of the nodes -- loss, accuracy, gradients equal, the
    per-node outputs equal after unpermute() -- while the 8-way halo shrinks from all-gather to a few %."""
    from han_amd import ops, reorder, synth
    from han_amd.gat import HeteGAT_multi
    n, f, p = 20000, 16, 2
    gen = torch.Generator(device=dev).manual_seed(3)
    shuf = torch.randperm(n, generator=gen, device=dev)
    graphs = [reorder.permute_graph(synth.banded_graph(n, 12, 150, 5 + q, dev), shuf) for q in range(p)]
    x = torch.randn((n, f), device=dev, generator=gen)
    labels = torch.randint(0, 3, (n,), device=dev, generator=gen, dtype=torch.int32)
    mask = (torch.rand((n,), device=dev, generator=gen) < 0.3).to(torch.uint8)
    wl = dict(x=x, labels=labels, train_mask=mask, val_mask=mask, graphs=graphs)
    rel = reorder.relabel(wl)
    assert max(reorder.halo_fraction(rel.wl["graphs"], 8)) < 0.15 < 0.6 < min(reorder.halo_fraction(graphs, 8))
    res = []
    for w, back in ((wl, lambda t: t), (rel.wl, rel.unpermute)):
        model = HeteGAT_multi().build(p, f, 3, device=dev, generator=torch.Generator().manual_seed(0))
        model.zero_grad_flat()
        M = model.node_level([w["x"]] * p, w["graphs"], 0.0, 0.0, True, ops.ACT_ELU)
        Z, att = model.semantic(M)
        loss, acc, logits = model.classifier_loss(Z, w["labels"], w["train_mask"], 1.0 / int(mask.sum()))
        loss.backward()
        res.append((float(loss.detach()), float(acc), back(logits.detach()), back(Z.detach()), model.flat_grad.clone()))
    (l0, a0, lg0, z0, g0), (l1, a1, lg1, z1, g1) = res
    assert abs(l0 - l1) < 1e-5 and abs(a0 - a1) < 1e-6
    assert float((lg0 - lg1).abs().max()) < TOL and float((z0 - z1).abs().max()) < TOL
    assert float((g0 - g1).abs().max()) < 1e-4 * max(1.0, float(g0.abs().max()))


@pytest.mark.parametrize("c", [30, 100])
def test_many_classes_through_the_model(dev, c):
    """nb_classes = 30 (models/gat.py:68 leaves it free; the class-per-lane kernel) and 100 (above the
    classifier kernels: torch on the GPU): inference and loss + gradients against the oracle."""
    n, f, p = 80, 12, 2
    prob = make_problem(321, n, f, p, c, [0.06, 0.4])
    model, bp = build_model(prob, dev)
    lg, fe, att = ho.hetegat_multi_inference([prob["x"]] * p, c, n, False, 0.0, 0.0, prob["biases"], [8], [8, 1],
                                             prob["params"])
    x, graphs = gpu_inputs(prob, dev)
    with torch.no_grad():
        logits, final_embed, _ = model.inference([x] * p, c, n, False, 0.0, 0.0, graphs, [8], [8, 1])
    assert np.abs(logits[0].cpu().numpy() - lg[0]).max() < TOL
    loss_ref, gref, lg_ref = _oracle_grads(prob, bp, dense=False)
    loss, grads, lgg, acc = _gpu_loss_and_grads(model, prob, dev)
    assert abs(loss - loss_ref) < 5e-4 and np.abs(lgg - lg_ref).max() < 5 * TOL
    acc_ref = float(ht.masked_accuracy(torch.tensor(lg_ref), torch.tensor(prob["onehot"]), torch.tensor(prob["mask"])))
    assert abs(acc - acc_ref) < 1e-5
    for k in ht.PARAM_ORDER:
        assert rel_err(grads[k], gref[k]) < GTOL, k


def test_structural_properties_of_the_node_attention(dev):
    """Properties SURVEY.md section 4 lists, on the real kernels (no oracle needed):
    (1) the order of the stored neighbours inside a row does not matter;
    (2) K single-head calls of the reference-named attn_head, concatenated, equal the K-head model layer
        (models/gat.py:42-46 builds the layer exactly that way);
    (3) a repeated neighbour is a repeated softmax term (the multigraph reading of a CSR with repeated entries):
        doubling every edge of a row changes nothing, doubling ONE edge shifts weight to it;
    (4) rows are independent: changing the neighbour list of row r changes only output row r."""
    import torch.nn.functional as Fnn
    from han_amd import layers, ops
    from han_amd.graph import CSRGraph
    prob = make_problem(77, 120, 10, 1, 3, [0.08])
    model, bp = build_model(prob, dev)
    x, graphs = gpu_inputs(prob, dev)
    g = graphs[0]
    with torch.no_grad():
        M = model.node_level([x], [g], 0.0, 0.0, False, ops.ACT_ELU)[:, 0, :]
        # (1) shuffle the columns inside every row
        gen = torch.Generator(device=dev).manual_seed(1)
        rows = torch.repeat_interleave(torch.arange(g.n_rows, device=dev), g.degrees())
        key = rows.double() + torch.rand(g.nnz, device=dev, generator=gen, dtype=torch.float64) * 0.5
        perm = torch.sort(key).indices
        gs = CSRGraph(g.rowptr, g.colidx[perm].contiguous(), g.n_cols)
        assert not torch.equal(gs.colidx, g.colidx)
        Ms = model.node_level([x], [gs], 0.0, 0.0, False, ops.ACT_ELU)[:, 0, :]
        assert float((M - Ms).abs().max()) < 2e-6
        # (2) eight single-head calls
        heads = []
        for k in range(8):
            params = {"W": model.W[0][:, 8 * k:8 * k + 8].contiguous(), "a1": model.a1[0, k], "b1": model.b1[0, k],
                      "a2": model.a2[0, k], "b2": model.b2[0, k], "c": model.c[0][8 * k:8 * k + 8].contiguous()}
            heads.append(layers.attn_head(x[None], 8, g, Fnn.elu, params=params)[0])
        assert float((torch.cat(heads, 1) - M).abs().max()) < 2e-6
        # (3) every edge doubled: identical; one extra copy of a single edge: only that row moves
        dbl = CSRGraph(g.rowptr * 2, torch.repeat_interleave(g.colidx, 2).contiguous(), g.n_cols)
        Md = model.node_level([x], [dbl], 0.0, 0.0, False, ops.ACT_ELU)[:, 0, :]
        assert float((M - Md).abs().max()) < 2e-6
        r = 17
        s, e = int(g.rowptr[r]), int(g.rowptr[r + 1])
        cols = torch.cat([g.colidx[:e], g.colidx[s:s + 1], g.colidx[e:]])
        rp = g.rowptr.clone()
        rp[r + 1:] += 1
        one = CSRGraph(rp, cols.contiguous(), g.n_cols)
        Mo = model.node_level([x], [one], 0.0, 0.0, False, ops.ACT_ELU)[:, 0, :]
        # (4) only row r changed
        changed = ((M - Mo).abs().max(1).values > 1e-7).nonzero().flatten().tolist()
        assert changed == [r] or (changed == [] and e - s == 1)
