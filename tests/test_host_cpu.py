"""CPU tests of the host logic above the C ABI (no GPU, no HIP compute): graph
containers, partitioning, autograd wiring, flat buffers, trainer -- with the
kernels replaced by tests/cpu_backend.py -- and the product's loud failures."""
import math

import numpy as np
import pytest
import torch

from oracle import han_oracle as ho
from oracle import han_oracle_torch as ht
from tests import rng_ref
from tests.helpers import group_masks, load_params, make_problem, rel_err


@pytest.fixture()
def cpu_ops(monkeypatch):
    from han_amd import ops
    from tests import cpu_backend
    for n in cpu_backend._NAMES:
        monkeypatch.setattr(ops, n, getattr(cpu_backend, n))
    return ops


def _cpu_model(prob):
    from han_amd.gat import HeteGAT_multi
    model = HeteGAT_multi().build(prob["p"], prob["f"], prob["c"], device="cpu")
    bp = ht.to_batched(prob["params"])
    load_params(model, bp)
    return model, bp


def _cpu_graphs(prob):
    from han_amd.graph import CSRGraph
    return [CSRGraph.from_bias(torch.tensor(b, dtype=torch.float32)) for b in prob["biases"]]


# --------------------------------------------------------------------------- graphs
def test_csr_from_bias_transpose_and_validation():
    from han_amd.graph import CSRGraph, as_graph
    prob = make_problem(3, 30, 4, 1, 3, [0.2])
    g = CSRGraph.from_bias(torch.tensor(prob["biases"][0]))
    rp, ci = ho.bias_to_csr(prob["biases"][0])
    assert np.array_equal(g.rowptr.numpy(), rp) and np.array_equal(g.colidx.numpy(), ci)
    t = g.transpose()
    import scipy.sparse as sp
    a = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(30, 30)).T.tocsr()
    a.sort_indices()
    assert np.array_equal(t.rowptr.numpy(), a.indptr) and np.array_equal(t.colidx.numpy(), a.indices)
    assert t.transpose() is g
    assert torch.equal(g.to_bias(torch.float64)[0], torch.tensor(prob["biases"][0][0]))
    assert as_graph((rp, ci)).nnz == g.nnz
    with pytest.raises(ValueError):
        CSRGraph(torch.tensor([0, 2]), torch.tensor([0, 1], dtype=torch.int32))       # rowptr dtype
    with pytest.raises(ValueError):
        CSRGraph.from_arrays([0, 2], [0, 7], 2)                                       # col out of range
    with pytest.raises(ValueError):
        CSRGraph.from_arrays([0, 3], [0, 1], 2)                                       # rowptr[-1] != nnz
    with pytest.raises(ValueError):
        CSRGraph.from_bias(torch.zeros(2, 3, 3))                                      # batch must be 1
    sp_t = torch.sparse_coo_tensor(torch.tensor([[0, 0, 0], [0, 1, 1], [1, 0, 1]]), torch.ones(3), (1, 2, 2))
    gs = as_graph(sp_t)
    assert gs.rowptr.tolist() == [0, 1, 3] and gs.colidx.tolist() == [1, 0, 1]


def test_lean_kernels_are_for_small_tables_with_long_rows():
    """ops._use_lean (HAN_FLAG_LEAN): fp32 tables that fit the L2s (<= 16384 rows) with long rows (mean degree >= 64),
    any id order."""
    from han_amd import ops
    from han_amd.graph import CSRGraph
    rp2 = torch.arange(0, 100 * 70 + 1, 70, dtype=torch.int64)
    long_rows = CSRGraph(rp2, torch.randint(0, 100, (7000,), dtype=torch.int32), 100)
    assert ops._use_lean(long_rows, torch.zeros((100, 64)))
    assert not ops._use_lean(long_rows, torch.zeros((100, 64), dtype=torch.bfloat16))
    short = CSRGraph(torch.tensor([0, 3, 3, 5, 6], dtype=torch.int64), torch.tensor([0, 2, 3, 1, 1, 0], dtype=torch.int32), 4)
    assert not ops._use_lean(short, torch.zeros((4, 64)))
    big = CSRGraph(torch.zeros(20001, dtype=torch.int64), torch.zeros(0, dtype=torch.int32), 20000)
    assert not ops._use_lean(big, torch.zeros((20000, 64)))


def test_partition_shards_cover_the_graph():
    from han_amd.dist import NodePartition
    prob = make_problem(4, 41, 4, 1, 3, [0.15])
    g = _cpu_graphs(prob)[0]
    gt = g.transpose()
    for world in (1, 2, 3, 8):
        rows_seen, cols_seen = [], []
        for r in range(world):
            part = NodePartition(41, r, world)
            assert part.n_table == part.shard * world >= 41
            rl, cl = part.shard_graph(g)
            assert rl.n_rows == part.n_local == cl.n_rows and rl.n_cols == part.n_table
            rows_seen.append(rl.colidx)
            cols_seen.append(cl.colidx)
            assert torch.equal(rl.rowptr, g.rowptr[part.row_start:part.row_end + 1] - g.rowptr[part.row_start])
        assert torch.equal(torch.cat(rows_seen), g.colidx)
        assert torch.equal(torch.cat(cols_seen), gt.colidx)
    with pytest.raises(ValueError):
        NodePartition(10, 3, 2)


def test_csr_graph_properties_randomised():
    """CSRGraph invariants on random graphs: transpose is an involution that keeps the edge
    multiset (and permutes edge values with it), row_split chunks tile exactly the long rows,
    to_bias / from_bias round-trip."""
    from han_amd.graph import CSRGraph
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = int(rng.integers(1, 60))
        deg = rng.integers(0, 9, size=n)
        if trial % 3 == 0:
            deg[rng.integers(0, n)] = 40            # one long row
        rp = np.concatenate([[0], np.cumsum(deg)])
        ci = np.concatenate([np.sort(rng.choice(n, size=d, replace=d > n)) for d in deg] + [np.zeros(0, int)])
        vals = torch.tensor(rng.standard_normal(len(ci)), dtype=torch.float32)
        g = CSRGraph(torch.tensor(rp), torch.tensor(ci, dtype=torch.int32), n, values=vals)
        t = g.transpose()
        assert t.n_rows == n and t.nnz == g.nnz
        rows = np.repeat(np.arange(n), deg)
        trows = np.repeat(np.arange(n), t.degrees().numpy())
        fwd = sorted(zip(rows.tolist(), ci.tolist(), vals.tolist()))
        bwd = sorted(zip(t.colidx.tolist(), trows.tolist(), t.values.tolist()))
        assert fwd == bwd
        assert t.transpose() is g
        sp = g.row_split(split_deg=16, chunk=8)
        long_rows = np.nonzero(deg > 16)[0]
        if len(long_rows) == 0:
            assert sp is None
        else:
            assert sp["long_rows"].tolist() == long_rows.tolist()
            covered = []
            for c in range(sp["n_chunks"]):
                r = int(sp["long_rows"][int(sp["chunk_long"][c])])
                s0, e0 = int(sp["chunk_start"][c]), int(sp["chunk_end"][c])
                assert rp[r] <= s0 < e0 <= rp[r + 1] and e0 - s0 <= 8
                covered += list(range(s0, e0))
            want = [e for r in long_rows for e in range(rp[r], rp[r + 1])]
            assert covered == want
        if g.nnz and len(set(zip(rows.tolist(), ci.tolist()))) == g.nnz:      # no duplicate entries
            g2 = CSRGraph.from_bias(g.to_bias())
            assert g2.rowptr.tolist() == list(rp) and g2.colidx.tolist() == list(ci)


# ------------------------------------------------------------- backward derivation
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_cpu_backend_gradients_match_oracle(cpu_ops, drop):
    """The hand-derived backward + the autograd wiring + the flat
    gradient buffer, against float64 autograd of the oracle."""
    from han_amd import layers, rng as hrng
    prob = make_problem(21, 60, 10, 2, 3, [0.08, 0.5])
    model, bp = _cpu_model(prob)
    graphs = _cpu_graphs(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    hrng.manual_seed(5)
    seeds = [hrng.next_seed() for _ in range(2)]
    hrng.manual_seed(5)
    model.zero_grad_flat()
    M = model.node_level([x, x], graphs, drop, drop, True, 1)
    Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
    labels = torch.tensor(prob["labels"], dtype=torch.int32)
    mask = torch.tensor(prob["mask"].astype(np.uint8))
    loss, acc, logits = layers.ClassifierLoss.apply(Z, model.Wc, model.bc, labels, mask,
                                                    1.0 / int(prob["mask"].sum()))
    loss.backward()
    masks = None
    keep = 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = []
        for q in range(2):
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            masks.append({"seq": torch.tensor(rng_ref.seq_mask(seeds[q], 60, 10, 8, drop)),
                          "coef": torch.tensor(rng_ref.coef_mask_csr(seeds[q], rp, ci, 8, drop)),
                          "fts": torch.tensor(rng_ref.fts_mask(seeds[q], 60, 64, drop))})
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    lg_ref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * 2, og, bpo, keep_in=keep,
                                      keep_coef=keep, masks=masks)
    loss_ref = ht.masked_softmax_cross_entropy(lg_ref, torch.tensor(prob["onehot"]),
                                               torch.tensor(prob["mask"]))
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    off = 0
    for k in ht.PARAM_ORDER:
        g = getattr(model, k).grad
        assert rel_err(g.numpy(), bpo[k].grad.numpy()) < 1e-4, k
        n = g.numel()                       # .grad is a view of the flat buffer, in PARAM order
        assert g.data_ptr() == model.flat_grad[off:off + n].data_ptr()
        off += n
    assert off == model.flat.numel()


def test_weighted_adjacency_values_on_cpu_backend(cpu_ops):
    """sp_attn_head's stored values scale the logits (layers.py:95-98): the host path
    (CSRGraph.values, their permutation into the transposed graph, the autograd
    wiring) against float64 autograd of the CSR oracle."""
    from han_amd import layers
    from han_amd.graph import CSRGraph
    n = 40
    prob = make_problem(3, n, 6, 1, 3, [0.2])
    rng = np.random.default_rng(0)
    rp, ci = ho.bias_to_csr(prob["biases"][0])
    vals = torch.tensor(rng.uniform(-1.5, 2.0, size=len(ci)), dtype=torch.float32)
    g = CSRGraph(torch.tensor(rp), torch.tensor(ci, dtype=torch.int32), n, values=vals)
    gt = g.transpose()
    dense = torch.zeros(n, n)
    dense[np.repeat(np.arange(n), np.diff(rp)), ci] = vals
    assert torch.equal(dense.t()[_rows(gt), gt.colidx.long()], gt.values)
    bp = ht.to_batched(prob["params"])
    names = ("W", "a1", "b1", "a2", "b2", "c")
    leaf = {k: bp[k][0].clone().to(torch.float32).requires_grad_(True) for k in names}
    cfg = {"train": True, "in_drop": 0.0, "coef_drop": 0.0, "seeds": (1,), "act": 1, "part": None}
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    M = layers.NodeLevelAttention.apply(None, *(leaf[k][None] for k in names), None, None, (x,), (g,), cfg)
    wgt = torch.tensor(rng.standard_normal((n, 64)))
    (M[:, 0, :] * wgt.to(torch.float32)).sum().backward()
    ref = {k: bp[k][0].clone().requires_grad_(True) for k in names}
    out = ht.node_attention_csr(torch.tensor(prob["x"][0]), torch.tensor(rp), torch.tensor(ci),
                                *(ref[k] for k in names), adj_vals=vals.to(torch.float64))
    (out * wgt).sum().backward()
    assert np.abs(M[:, 0, :].detach().numpy() - out.detach().numpy()).max() < 1e-5
    for k in names:
        assert rel_err(leaf[k].grad.numpy(), ref[k].grad.numpy()) < 1e-4, k


def _rows(g):
    return torch.repeat_interleave(torch.arange(g.n_rows), g.degrees())


def test_trainer_epochs_match_oracle_on_cpu_backend(cpu_ops):
    from han_amd.trainer import HANTrainer
    prob = make_problem(31, 50, 8, 2, 3, [0.1, 0.4])
    model, bp = _cpu_model(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    tr = HANTrainer(model, [x, x], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                    torch.tensor(prob["mask"].astype(np.uint8)), torch.tensor((~prob["mask"]).astype(np.uint8)),
                    attn_drop=0.0, ffd_drop=0.0)
    bpo = {k: v.clone() for k, v in bp.items()}
    st = ht.new_adam_state(bpo)
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    for _ in range(3):
        tl, ta, vl, va = tr.epoch()
        _, vloss, vacc = ht.train_epoch([torch.tensor(prob["x"][0])] * 2, og, bpo, st,
                                        torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]),
                                        torch.tensor(~prob["mask"]), keep=1.0)
        assert abs(float(vl) - vloss) < 1e-5 and abs(float(va) - vacc) < 1e-6
    for k in ht.PARAM_ORDER:
        assert np.abs(getattr(model, k).detach().numpy() - bpo[k].numpy()).max() < 1e-5, k
    # early-stopping bookkeeping of ex_acm3025.py:225-240
    assert tr.early_stopping(1.0, 0.5) is False and tr.best_state is not None
    tr.patience = 2
    assert tr.early_stopping(2.0, 0.1) is False
    assert tr.early_stopping(2.0, 0.1) is True


def test_device_step_state_mode_matches_oracle(cpu_ops):
    """HANTrainer(use_graph=True) keeps the per-step seed word and Adam's step count in a
    device state that every epoch advances (han_hip.h "Seeds"); on the CPU backend the same
    flow runs eagerly.  Masks rebuilt from resolve_seed(fixed seed, state) + the oracle
    loop must give the same parameters; two epochs must draw different masks."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    n, f, drop = 50, 8, 0.6
    prob = make_problem(31, n, f, 2, 3, [0.1, 0.4])
    model, bp = _cpu_model(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    hrng.manual_seed(123)
    tr = HANTrainer(model, [x, x], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                    torch.tensor(prob["mask"].astype(np.uint8)), torch.tensor((~prob["mask"]).astype(np.uint8)),
                    attn_drop=drop, ffd_drop=drop, use_graph=True)
    bpo = {k: v.clone() for k, v in bp.items()}
    st = ht.new_adam_state(bpo)
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    keep = rng_ref.keep_prob32(drop)
    seen = []
    for ep in range(1, 4):
        tl, ta, vl, va = tr.epoch()
        assert int(tr.step_state[1]) == ep == tr.opt.t
        word = int(tr.step_state[0]) & ((1 << 64) - 1)
        assert word == (ep * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        fixed = model._fixed_seeds[(0, 2)]
        masks = []
        for q in range(2):
            sd = rng_ref.resolve_seed(fixed[q], word)
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            masks.append({"seq": torch.tensor(rng_ref.seq_mask(sd, n, f, 8, drop)),
                          "coef": torch.tensor(rng_ref.coef_mask_csr(sd, rp, ci, 8, drop)),
                          "fts": torch.tensor(rng_ref.fts_mask(sd, n, 64, drop))})
        seen.append(masks[0]["fts"])
        _, vloss, vacc = ht.train_epoch([torch.tensor(prob["x"][0])] * 2, og, bpo, st,
                                        torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]),
                                        torch.tensor(~prob["mask"]), keep=keep, masks=masks)
        assert abs(float(vl) - vloss) < 2e-5 and abs(float(va) - vacc) < 1e-6
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    for k in ht.PARAM_ORDER:
        assert np.abs(getattr(model, k).detach().numpy() - bpo[k].numpy()).max() < 2e-5, k


def test_eval_forward_does_not_write_gradients(cpu_ops):
    """An eval forward (torch.no_grad()) leaves the flat gradient buffer alone: the fused classifier computes and, in
    direct-gradient mode, WRITES dWc / dbc only when the caller's grad mode is on (layers.classifier_loss_any)."""
    from han_amd import layers
    from han_amd.trainer import HANTrainer
    prob = make_problem(33, 40, 8, 2, 3, [0.1, 0.4])
    model, _ = _cpu_model(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    tr = HANTrainer(model, [x, x], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                    torch.tensor(prob["mask"].astype(np.uint8)), attn_drop=0.0, ffd_drop=0.0)
    tr.train_step()
    marker = torch.full_like(model.flat_grad, 7.0)
    model.flat_grad.copy_(marker)
    tr.eval_step()
    assert torch.equal(model.flat_grad, marker)
    Z = torch.randn(40, 64)
    layers.classifier_loss_any(Z, model.Wc, model.bc, tr.labels, tr.train_mask, tr.w_train)
    assert not torch.equal(model.flat_grad, marker)


def test_overlap_eval_returns_the_plain_epoch_numbers_one_call_late(cpu_ops):
    """HANTrainer(use_graph=True, overlap_eval=True): call k returns the training pair of step k and the validation
    pair of the parameters BEFORE step k; flush_eval() gives the pair of the last epoch; early_stopping checkpoints
    the parameters its pair belongs to.  On the CPU backend the reordered body runs eagerly."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    n, f, drop = 50, 8, 0.6
    prob = make_problem(31, n, f, 2, 3, [0.1, 0.4])
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    hist, trainers = [], []
    for overlap in (False, True, "segments"):
        model, _ = _cpu_model(prob)
        hrng.manual_seed(123)
        tr = HANTrainer(model, [x, x], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                        torch.tensor(prob["mask"].astype(np.uint8)), torch.tensor((~prob["mask"]).astype(np.uint8)),
                        attn_drop=drop, ffd_drop=drop, use_graph=True, overlap_eval=bool(overlap))
        # "segments": the flow of the CAPTURED epoch -- the eval forward in two pieces inside the training step's
        # forward and backward (layers.NodeLevelAttention) -- run in place
        tr._force_segments = overlap == "segments"
        if overlap is True:
            init = tr.model.flat.detach().clone()
        hist.append([[float(v) for v in tr.epoch()] for _ in range(4)])
        trainers.append(tr)
    plain, over, seg = hist
    for k in range(4):
        assert over[k][:2] == plain[k][:2], k                     # the training step is untouched
        assert seg[k] == over[k], (k, seg[k], over[k])
        if k:
            assert over[k][2:] == plain[k - 1][2:], k             # validation pair of the epoch before
    assert torch.equal(trainers[2].model.flat, trainers[0].model.flat)
    tr = trainers[1]
    assert torch.equal(tr._flat_prev, trainers[0].model.flat) is False      # (it holds the parameters before step 4)
    tr.early_stopping(over[3][2], over[3][3])
    assert tr.best_state is not None and not torch.equal(tr.best_state, tr.model.flat)
    assert [float(v) for v in tr.flush_eval()] == plain[3][2:]
    assert torch.equal(tr._flat_prev, tr.model.flat) and torch.equal(tr.model.flat, trainers[0].model.flat)
    # the first call evaluates the initial parameters
    model0, _ = _cpu_model(prob)
    assert torch.equal(model0.flat.detach(), init)
    with pytest.raises(ValueError):
        HANTrainer(model0, [x, x], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                   torch.tensor(prob["mask"].astype(np.uint8)), overlap_eval=True)


# -------------------------------------------------------------------- model surface
def test_model_variables_and_initialisers():
    from han_amd.gat import HeteGAT_multi
    g = torch.Generator().manual_seed(0)
    m = HeteGAT_multi().build(2, 1870, 3, (8,), (8, 1), 128, device="cpu", generator=g)
    shapes = {k: tuple(getattr(m, k).shape) for k in ht.PARAM_ORDER}
    assert shapes == {"W": (2, 1870, 64), "a1": (2, 8, 8), "b1": (2, 8), "a2": (2, 8, 8), "b2": (2, 8),
                      "c": (2, 64), "w_omega": (64, 128), "b_omega": (128,), "u_omega": (128,),
                      "Wc": (1, 64, 3), "bc": (1, 3)}
    # 248,419 trainables at F=1870
    assert m.flat.numel() == 16 * (1870 * 8 + 8 + 1 + 8 + 1 + 8) + (64 * 128 + 128 + 128) + (64 * 3 + 3)
    assert float(m.W.abs().max()) <= math.sqrt(6.0 / (1870 + 8)) + 1e-7      # glorot-uniform conv1d
    assert float(m.a1.abs().max()) <= math.sqrt(6.0 / 9) + 1e-7
    assert float(m.b1.abs().max()) == 0 and float(m.c.abs().max()) == 0 and float(m.bc.abs().max()) == 0
    assert abs(float(m.w_omega.std()) - 0.1) < 0.01
    m2 = HeteGAT_multi().build(2, 10, 3, (8, 8), (8, 8, 1), device="cpu")      # models/gat.py:48-57
    assert tuple(m2.W_1.shape) == (2, 64, 64) and tuple(m2.a1_1.shape) == (2, 8, 8)
    # widths other than 8 x 8 (models/gat.py:42-57 leaves hid_units / n_heads free): variables keep the
    # reference's shapes, whatever K*F' is; mp_att_size is free up to 128
    m3 = HeteGAT_multi().build(2, 10, 3, (16,), (8, 1), 48, device="cpu")       # K*F' = 128
    assert tuple(m3.W.shape) == (2, 10, 128) and tuple(m3.w_omega.shape) == (128, 48) and tuple(m3.Wc.shape) == (1, 128, 3)
    m4 = HeteGAT_multi().build(2, 10, 3, (8, 8), (12, 4, 1), device="cpu")      # 96 wide, then 32 wide
    assert tuple(m4.W_1.shape) == (2, 96, 32) and tuple(m4.w_omega.shape) == (32, 128)
    m5 = HeteGAT_multi().build(2, 10, 3, (10,), (4, 1), device="cpu")          # any head width up to 64
    assert tuple(m5.W.shape) == (2, 10, 40) and tuple(m5.a1.shape) == (2, 4, 10)
    m7 = HeteGAT_multi().build(2, 10, 3, (65,), (2, 1), device="cpu")          # heads wider than the 64-column group: slices
    assert tuple(m7.W.shape) == (2, 10, 130) and tuple(m7.a1.shape) == (2, 2, 65) and tuple(m7.w_omega.shape) == (130, 128)
    with pytest.raises(ValueError):
        HeteGAT_multi().build(2, 10, 3, (0,), (1, 1), device="cpu")
    m6 = HeteGAT_multi().build(2, 10, 3, (32,), (8, 1), device="cpu")          # final width 256: run-time-width K3 / classifier kernels
    assert tuple(m6.w_omega.shape) == (256, 128) and tuple(m6.Wc.shape) == (1, 256, 3)
    assert tuple(HeteGAT_multi().build(2, 10, 40, device="cpu").Wc.shape) == (1, 64, 40)   # up to 64 classes
    assert tuple(HeteGAT_multi().build(2, 10, 100, mp_att_size=200, device="cpu").w_omega.shape) == (64, 200)
    with pytest.raises(ValueError):
        HeteGAT_multi().build(2, 10, 3, (8, 8), (8, 1), device="cpu")          # n_heads too short


def test_no_cpu_fallback_in_the_product():
    """The product path must fail loudly on CPU tensors (no silent fallback)."""
    from han_amd import layers, ops
    from han_amd.gat import HeteGAT_multi
    x = torch.zeros(4, 2, 64)
    with pytest.raises(ValueError, match="GPU"):
        ops.sem_attn_fwd(x, torch.zeros(64, 128), torch.zeros(128), torch.zeros(128))
    with pytest.raises(ValueError, match="GPU"):
        ops.project_fwd(torch.zeros(4, 5), torch.zeros(5, 64), torch.zeros(8, 8), torch.zeros(8, 8),
                        torch.zeros(8), torch.zeros(8))
    HeteGAT_multi.reset_default()
    with pytest.raises(ValueError, match="GPU"):
        HeteGAT_multi.inference([torch.zeros(1, 4, 5)], 3, 4, False, 0.0, 0.0, [torch.zeros(1, 4, 4)],
                                [8], [8, 1])                                   # class-level call, as the reference
    HeteGAT_multi.reset_default()
    with pytest.raises(ValueError):
        layers._squeeze_batch(torch.zeros(2, 4, 5))


def test_rng_seed_stream_is_deterministic():
    from han_amd import rng
    rng.manual_seed(1)
    a = [rng.next_seed() for _ in range(4)]
    rng.manual_seed(1)
    assert a == [rng.next_seed() for _ in range(4)] and len(set(a)) == 4
    m = rng_ref.seq_mask(a[0], 64, 32, 8, 0.6)
    assert abs(m.mean() - 0.4) < 0.02 and not np.array_equal(m[0], m[1])
    assert np.array_equal(m, rng_ref.seq_mask(a[0], 64, 32, 8, 0.6))
    # global-id keys: a row block of a partition draws the same bits
    assert np.array_equal(rng_ref.fts_mask(a[1], 10, 64, 0.6, row_offset=7),
                          rng_ref.fts_mask(a[1], 17, 64, 0.6)[7:])


def test_base_gattn_loss_functions():
    from han_amd.base_gattn import BaseGAttN
    rng = np.random.default_rng(2)
    logits, labels = rng.standard_normal((30, 4)), np.eye(4)[rng.integers(0, 4, 30)]
    mask = rng.random(30) < 0.5
    a = float(BaseGAttN.masked_softmax_cross_entropy(torch.tensor(logits), torch.tensor(labels), torch.tensor(mask)))
    assert abs(a - ho.masked_softmax_cross_entropy(logits, labels, mask)) < 1e-12
    b = float(BaseGAttN.masked_accuracy(torch.tensor(logits), torch.tensor(labels), torch.tensor(mask)))
    assert abs(b - ho.masked_accuracy(logits, labels, mask)) < 1e-12


def test_base_gattn_remaining_helpers():
    """The BaseGAttN members the HAN script never calls (models/base_gattn.py:5-10,26-35,50-59,71-94):
    class-weighted loss, preshape, confusion matrix, multi-label sigmoid loss, micro-F1 -- against the
    NumPy restatements."""
    from han_amd.base_gattn import BaseGAttN
    rng = np.random.default_rng(4)
    n, c = 57, 5
    logits = rng.standard_normal((n, c)) * 2
    labels = rng.integers(0, c, n)
    multi = (rng.random((n, c)) < 0.3).astype(np.int64)
    mask = rng.random(n) < 0.5
    cw = rng.uniform(0.5, 2.0, c)
    tl, tlab, tm = torch.tensor(logits), torch.tensor(labels), torch.tensor(mask)
    assert abs(float(BaseGAttN.loss(tl, tlab, c, torch.tensor(cw))) - ho.weighted_loss(logits, labels, c, cw)) < 1e-12
    a, b = BaseGAttN.preshape(tl.reshape(1, n, c), tlab.reshape(1, n), c)
    assert tuple(a.shape) == (n, c) and tuple(b.shape) == (n,)
    assert np.array_equal(BaseGAttN.confmat(tl, tlab).numpy(), ho.confmat(logits, labels))
    assert abs(float(BaseGAttN.masked_sigmoid_cross_entropy(tl, torch.tensor(multi), tm))
               - ho.masked_sigmoid_cross_entropy(logits, multi, mask)) < 1e-12
    assert abs(float(BaseGAttN.micro_f1(tl, torch.tensor(multi), tm)) - float(ho.micro_f1(logits, multi, mask))) < 1e-6


def test_evaluate_matches_the_reference_jhyexp_outputs():
    """tests/golden/jhyexp_ref.npz holds what the reference's OWN jhyexp.my_KNN / my_Kmeans
    (jhyexp.py:20-86, imported by tests/golden/gen_fixtures.py) printed / returned on seeded
    embeddings; han_amd.evaluate must reproduce them from the same seed (KNN to the 4
    printed decimals, KMeans to rounding -- same scikit-learn, same random stream)."""
    import os
    import sklearn
    from han_amd import evaluate
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "jhyexp_ref.npz"))
    if str(z["sklearn_version"]) != sklearn.__version__:
        pytest.skip("fixture was generated with another scikit-learn")
    x, y, seed, time = z["x"], z["y"], int(z["seed"]), int(z["time"])
    knn = evaluate.my_KNN(x, np.eye(int(z["k_means"]))[y], k=int(z["k_knn"]), time=time, seed=seed, verbose=False)
    for split, macro, micro in z["knn"]:
        got = knn[float(split)]
        assert abs(got[0] - macro) < 6e-5 and abs(got[1] - micro) < 6e-5, (split, got, macro, micro)
    nmi, ari = evaluate.my_Kmeans(x, y, k=int(z["k_means"]), time=time, seed=seed, verbose=False)
    assert abs(nmi - z["kmeans"][0]) < 1e-12 and abs(ari - z["kmeans"][1]) < 1e-12


def test_evaluate_and_checkpoint(cpu_ops, tmp_path):
    from han_amd import evaluate
    from han_amd.trainer import HANTrainer
    rng = np.random.default_rng(0)
    y = rng.integers(0, 3, 300)
    x = np.eye(3)[y] * 3 + rng.standard_normal((300, 3)) * 0.3      # separable embeddings
    knn = evaluate.my_KNN(x, np.eye(3)[y], time=2, seed=0, verbose=False)
    assert set(knn) == {0.2, 0.4, 0.6, 0.8} and all(v[0] > 0.95 and v[1] > 0.95 for v in knn.values())
    nmi, ari = evaluate.my_Kmeans(x, y, k=3, time=2, seed=0, verbose=False)
    assert nmi > 0.9 and ari > 0.9
    prob = make_problem(41, 30, 6, 1, 3, [0.2])
    model, _ = _cpu_model(prob)
    xt = torch.tensor(prob["x"][0], dtype=torch.float32)
    tr = HANTrainer(model, [xt], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                    torch.tensor(prob["mask"].astype(np.uint8)), attn_drop=0.0, ffd_drop=0.0)
    tr.epoch()
    path = str(tmp_path / "ck.pt")
    tr.save_checkpoint(path)
    snap, t = model.flat.clone(), tr.opt.t
    tr.epoch()
    assert not torch.equal(snap, model.flat)
    tr.load_checkpoint(path)
    assert torch.equal(snap, model.flat) and tr.opt.t == t
    # device-step-state mode: the checkpointed Adam step count is pushed back to the device word,
    # so the epoch after a restore repeats the bias correction of the epoch after the save
    model2, _ = _cpu_model(prob)
    tr2 = HANTrainer(model2, [xt], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                     torch.tensor(prob["mask"].astype(np.uint8)), attn_drop=0.0, ffd_drop=0.0, use_graph=True)
    tr2.epoch(); tr2.epoch()
    tr2.save_checkpoint(path)
    tr2.epoch()
    after = model2.flat.clone()
    tr2.epoch()
    tr2.load_checkpoint(path)
    assert int(tr2.step_state[1]) == 2 == tr2.opt.t
    tr2.epoch()
    assert int(tr2.step_state[1]) == 3 and torch.allclose(after, model2.flat, atol=0, rtol=0)


@pytest.mark.parametrize("drop,residual", [(0.0, False), (0.6, False), (0.0, True), (0.6, True)])
def test_multi_layer_stack_on_cpu_backend(cpu_ops, drop, residual):
    """models/gat.py:48-57: hid_units=[8,16], n_heads=[8,4,1] -- the second layer
    reads the first layer's concatenated heads; gradients flow back through both."""
    from han_amd import layers, rng as hrng
    from han_amd.gat import HeteGAT_multi
    prob = make_problem(61, 40, 9, 2, 3, [0.15, 0.4], hid_units=[8, 16], n_heads=(8, 4, 1),
                        residual=residual)
    bp = ht.to_batched(prob["params"])
    assert ("Wr_1" in bp) == residual
    assert ht.n_extra_layers(bp) == 1 and bp["W_1"].shape == (2, 64, 64) and bp["a1_1"].shape == (2, 4, 16)
    model = HeteGAT_multi().build(2, 9, 3, (8, 16), (8, 4, 1), device="cpu", residual=residual)
    load_params(model, bp)
    graphs = _cpu_graphs(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    hrng.manual_seed(9)
    seeds = [hrng.next_seed() for _ in range(4)]          # layer 0: p0,p1 ; layer 1: p0,p1
    hrng.manual_seed(9)
    model.zero_grad_flat()
    M = model.node_level([x, x], graphs, drop, drop, True, 1)
    Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
    loss, _, _ = layers.ClassifierLoss.apply(Z, model.Wc, model.bc, torch.tensor(prob["labels"], dtype=torch.int32),
                                             torch.tensor(prob["mask"].astype(np.uint8)),
                                             1.0 / int(prob["mask"].sum()))
    loss.backward()
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = []
        for q in range(2):
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            mk = lambda sd, f, K: {"seq": torch.tensor(rng_ref.seq_mask(sd, 40, f, K, drop)),
                                   "coef": torch.tensor(rng_ref.coef_mask_csr(sd, rp, ci, K, drop)),
                                   "fts": torch.tensor(rng_ref.fts_mask(sd, 40, 64, drop))}
            m0 = mk(seeds[q], 9, 8)
            m0["layers"] = [mk(seeds[2 + q], 64, 4)]
            masks.append(m0)
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    lg_ref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * 2, og, bpo, keep_in=keep,
                                      keep_coef=keep, masks=masks)
    loss_ref = ht.masked_softmax_cross_entropy(lg_ref, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5
    for k in ht.param_order(bp):
        assert rel_err(getattr(model, k).grad.numpy(), bpo[k].grad.numpy()) < 1e-4, k
    # numpy restatement of the multi-layer forward (models/gat.py:48-57) agrees too
    lg_np, _, _ = ho.hetegat_multi_inference([prob["x"]] * 2, 3, 40, False, 0.0, 0.0, prob["biases"],
                                             [8, 16], [8, 4, 1], prob["params"])
    if drop == 0 and not residual:
        assert np.abs(lg_np[0] - lg_ref.detach().numpy()).max() < 1e-10


@pytest.mark.parametrize("drop,residual", [(0.0, False), (0.6, True)])
def test_wide_head_in_a_deeper_layer_on_cpu_backend(cpu_ops, drop, residual):
    """Host logic of layers.WideHeadAttention inside a stack (hid_units=[8,96], n_heads=[8,2,1]): slices of one
    head with shared scores / draws, totals of df1 / df2 across the slices, the input gradient into layer 0."""
    from han_amd import ops, rng as hrng
    from han_amd.gat import HeteGAT_multi
    n = 40
    prob = make_problem(64, n, 9, 2, 3, [0.15, 0.4], hid_units=[8, 96], n_heads=(8, 2, 1), residual=residual)
    bp = ht.to_batched(prob["params"])
    model = HeteGAT_multi().build(2, 9, 3, (8, 96), (8, 2, 1), device="cpu", residual=residual)
    load_params(model, bp)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    hrng.manual_seed(9)
    seeds = [hrng.next_seed() for _ in range(4)]
    hrng.manual_seed(9)
    model.zero_grad_flat()
    M = model.node_level([x, x], _cpu_graphs(prob), drop, drop, True, ops.ACT_ELU)
    assert tuple(M.shape) == (n, 2, 192)
    Z, _ = model.semantic(M)
    loss, _, logits = model.classifier_loss(Z, torch.tensor(prob["labels"], dtype=torch.int32),
                                            torch.tensor(prob["mask"].astype(np.uint8)), 1.0 / int(prob["mask"].sum()))
    loss.backward()
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = []
        for q in range(2):
            rp, ci = ho.bias_to_csr(prob["biases"][q])
            m0 = group_masks(seeds[q], n, 9, 8, 8, rp, ci, drop)
            m0["layers"] = [group_masks(seeds[2 + q], n, 64, 2, 96, rp, ci, drop)]
            masks.append(m0)
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    lg_ref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * 2, og, bpo, keep_in=keep, keep_coef=keep, masks=masks)
    loss_ref = ht.masked_softmax_cross_entropy(lg_ref, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    assert np.abs(logits.detach().numpy() - lg_ref.detach().numpy()).max() < 1e-4
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5
    for k in ht.param_order(bp):
        assert rel_err(getattr(model, k).grad.numpy(), bpo[k].grad.numpy()) < 1e-4, k


def test_mat_loader_follows_the_reference_script(tmp_path):
    """han_amd.process.load_data_mat == ex_acm3025.py:57-87 on a synthetic .mat with the
    ACM3025 keys: the '- I' on the meta-path matrices, masks, zeroed label rows, and
    three feature copies for two meta-paths."""
    import scipy.io as sio
    from han_amd import process
    rng = np.random.default_rng(0)
    n, f, c = 30, 7, 3
    pap = (rng.random((n, n)) < 0.2).astype(float)
    pap = np.maximum(pap, pap.T)
    np.fill_diagonal(pap, 1.0)
    plp = np.eye(n) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    label = np.eye(c)[rng.integers(0, c, n)]
    idx = rng.permutation(n)
    path = str(tmp_path / "ACM_like.mat")
    sio.savemat(path, {"label": label, "feature": rng.random((n, f)), "PAP": pap, "PLP": plp,
                       "train_idx": idx[None, :10], "val_idx": idx[None, 10:15], "test_idx": idx[None, 15:]})
    adj, fea, y_tr, y_va, y_te, m_tr, m_va, m_te = process.load_data_mat(path)
    assert len(adj) == 2 and len(fea) == 3 and fea[0].shape == (n, f)
    assert np.array_equal(adj[0], pap - np.eye(n)) and np.array_equal(adj[1], plp - np.eye(n))
    assert m_tr.sum() == 10 and m_va.sum() == 5 and m_te.sum() == 15 and not (m_tr & m_va).any()
    assert np.array_equal(y_tr[m_tr], label[m_tr]) and not y_tr[~m_tr].any()
    # adj_to_bias re-adds I (ex_acm3025.py:118): the edge set equals the original matrix's
    g = process.adj_to_graph(adj[0], nhood=1)
    rp, ci = ho.bias_to_csr(ho.adj_to_bias(adj[0][None], [n], 1))
    assert np.array_equal(g.rowptr.numpy(), rp) and np.array_equal(g.colidx.numpy(), ci)
    assert g.nnz == int((pap > 0).sum())


def test_checkpoint_resume_is_bit_exact_with_dropout(cpu_ops, tmp_path):
    """A resumed run must continue the SAME dropout mask stream and early-stopping bookkeeping:
    save after epoch 1, run epochs 2-3; a fresh trainer that loads the checkpoint and runs two
    epochs must end on identical parameters (the checkpoint carries the seed-stream state)."""
    from han_amd import rng as hrng
    from han_amd.trainer import HANTrainer
    prob = make_problem(43, 40, 6, 2, 3, [0.1, 0.3])
    xt = torch.tensor(prob["x"][0], dtype=torch.float32)

    def mk():
        model, _ = _cpu_model(prob)
        return model, HANTrainer(model, [xt, xt], _cpu_graphs(prob), torch.tensor(prob["labels"], dtype=torch.int32),
                                 torch.tensor(prob["mask"].astype(np.uint8)), attn_drop=0.6, ffd_drop=0.6)
    hrng.manual_seed(5)
    model, tr = mk()
    tr.epoch()
    tr.early_stopping(1.25, 0.5)
    path = str(tmp_path / "resume.pt")
    tr.save_checkpoint(path)
    tr.epoch(); tr.epoch()
    want = model.flat.clone()
    hrng.manual_seed(12345)                  # a different process: the stream state comes from the file
    model2, tr2 = mk()
    tr2.load_checkpoint(path)
    assert (tr2.vlss_mn, tr2.vacc_mx, tr2.curr_step) == (1.25, 0.5, 0) and tr2.best_state is not None
    tr2.epoch(); tr2.epoch()
    assert torch.equal(model2.flat, want)
    # a checkpoint of another configuration is refused
    prob3 = make_problem(43, 40, 7, 2, 3, [0.1, 0.3])
    model3, _ = _cpu_model(prob3)
    tr3 = HANTrainer(model3, [torch.zeros(40, 7)] * 2, _cpu_graphs(prob3), torch.tensor(prob["labels"], dtype=torch.int32),
                     torch.tensor(prob["mask"].astype(np.uint8)))
    with pytest.raises(ValueError):
        tr3.load_checkpoint(path)


def test_module_apply_keeps_parameters_bound_to_flat():
    """nn.Module.to()/.float()/... must not detach the parameter views from the flat buffer that
    Adam updates (they are re-created on the moved buffer); a dtype change is refused."""
    from han_amd.gat import HeteGAT_multi
    model = HeteGAT_multi().build(2, 5, 3, device="cpu")
    w0 = model.W.detach().clone()
    model.to("cpu").float()
    assert torch.equal(model.W, w0)
    lo = model.flat.data_ptr()
    hi = lo + model.flat.numel() * 4
    for name, _ in model.param_shapes():
        v = getattr(model, name)
        assert lo <= v.data_ptr() < hi and lo <= v.grad.data_ptr() - model.flat_grad.data_ptr() + lo < hi, name
    model.flat.add_(1.0)                     # what the optimiser does
    assert torch.equal(model.W, w0 + 1.0)
    with pytest.raises(TypeError):
        model.double()


def test_arbitrary_activation_callable_on_cpu_backend(cpu_ops):
    """models/gat.py:36 takes any `activation`: callables other than ELU / identity are applied by torch
    on the kernels' pre-activation, per head (the last axis is the head's F' features)."""
    from han_amd import layers
    prob = make_problem(77, 30, 6, 2, 3, [0.1, 0.4])
    model, bp = _cpu_model(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    for tact, nact in ((torch.tanh, np.tanh),
                       (lambda t: torch.softmax(t, -1), lambda a: np.exp(a - a.max(-1, keepdims=True))
                        / np.exp(a - a.max(-1, keepdims=True)).sum(-1, keepdims=True))):
        _, fe_ref, _ = ho.hetegat_multi_inference([prob["x"]] * 2, 3, 30, False, 0.0, 0.0, prob["biases"], [8], [8, 1],
                                                  prob["params"], activation=nact)
        code, post = layers._act_code(tact)
        with torch.no_grad():
            M = model.node_level([x, x], _cpu_graphs(prob), 0.0, 0.0, False, code, post=post)
            Z, _ = layers.SemanticAttention.apply(M, model.w_omega, model.b_omega, model.u_omega)
        assert np.abs(Z.numpy() - fe_ref).max() < 1e-5


@pytest.mark.parametrize("K,FP,A", [(8, 16, 128), (4, 8, 128), (3, 8, 48), (12, 8, 80),
                                    (5, 12, 128), (3, 20, 64), (2, 3, 32), (1, 50, 100),
                                    (8, 32, 128), (3, 64, 40),         # 256 / 192 wide: above the K3 kernels
                                    (8, 8, 200),                       # attention size above the K3 kernels
                                    # heads wider than 64 columns: slices of one head (layers.WideHeadAttention)
                                    (2, 128, 64), (1, 100, 128), (3, 96, 40)])
@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_general_head_widths_on_cpu_backend(cpu_ops, K, FP, A, drop):
    """Host logic of the widths other than 8 x 8 (han_amd.gat.node_level / semantic / classifier_loss): head
    groups of 64 columns with seed + g, zero-weight completion heads, head widths that are not a lane-mapped
    size (12, 20, 3, 50: run at 16, 32, 4, 64 with zero-weight columns), zero-padded K3 / classifier operands --
    loss and every gradient against float64 autograd of the oracle with the same hash masks."""
    from han_amd import ops, rng as hrng
    from han_amd.gat import HeteGAT_multi
    n, f, p = 40, 9, 2
    prob = make_problem(900 + K, n, f, p, 3, [0.1, 0.4], hid_units=[FP], n_heads=(K, 1), mp_att_size=A)
    model = HeteGAT_multi().build(p, f, 3, (FP,), (K, 1), A, device="cpu")
    bp = ht.to_batched(prob["params"])
    load_params(model, bp)
    hrng.manual_seed(11)
    seeds = [hrng.next_seed() for _ in range(p)]
    hrng.manual_seed(11)
    masks, keep = None, 1.0
    if drop > 0:
        keep = rng_ref.keep_prob32(drop)
        masks = [group_masks(seeds[q], n, f, K, FP, *ho.bias_to_csr(prob["biases"][q]), drop) for q in range(p)]
    bpo = {k: v.clone().requires_grad_(True) for k, v in bp.items()}
    og = [tuple(torch.tensor(t) for t in ho.bias_to_csr(b)) for b in prob["biases"]]
    lref, _, _ = ht.hetegat_forward([torch.tensor(prob["x"][0])] * p, og, bpo, keep_in=keep, keep_coef=keep, masks=masks)
    loss_ref = ht.masked_softmax_cross_entropy(lref, torch.tensor(prob["onehot"]), torch.tensor(prob["mask"]))
    loss_ref.backward()
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    model.zero_grad_flat()
    M = model.node_level([x] * p, _cpu_graphs(prob), drop, drop, True, ops.ACT_ELU)
    assert tuple(M.shape) == (n, p, K * FP)
    Z, _ = model.semantic(M)
    loss, _, logits = model.classifier_loss(Z, torch.tensor(prob["labels"], dtype=torch.int32),
                                            torch.tensor(prob["mask"].astype(np.uint8)), 1.0 / int(prob["mask"].sum()))
    loss.backward()
    assert np.abs(logits.detach().numpy() - lref.detach().numpy()).max() < 1e-4
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-5
    for k in ht.PARAM_ORDER:
        assert rel_err(getattr(model, k).grad.numpy(), bpo[k].grad.numpy()) < 1e-4, k


def test_locality_pass_relabelling_roundtrip():
    """han_amd.reorder: permute_graph keeps the edge multiset (values travel with the edges), bfs_order is a
    permutation that recovers hidden locality (mean |i - j| and the 8-way halo shrink on a banded graph
    whose ids were shuffled), Relabelled permutes per-node tensors in and unpermute() brings outputs back."""
    from han_amd import reorder, synth
    from han_amd.graph import CSRGraph
    n = 1500
    g = synth.banded_graph(n, 8, 20, 3)
    vals = torch.arange(g.nnz, dtype=torch.float32)
    g = CSRGraph(g.rowptr, g.colidx, n, values=vals)
    shuf = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    gs = reorder.permute_graph(g, shuf)
    inv = torch.empty_like(shuf)
    inv[shuf] = torch.arange(n)
    rows = torch.repeat_interleave(torch.arange(n), g.degrees())
    old = sorted(zip(inv[rows].tolist(), inv[g.colidx.long()].tolist(), g.values.tolist()))
    rows2 = torch.repeat_interleave(torch.arange(n), gs.degrees())
    new = sorted(zip(rows2.tolist(), gs.colidx.tolist(), gs.values.tolist()))
    assert old == new

    def spread(gr):
        r = torch.repeat_interleave(torch.arange(n), gr.degrees())
        d = (gr.colidx.long() - r).abs()
        return float(torch.minimum(d, n - d).float().mean())
    perm = reorder.bfs_order([gs])
    assert sorted(perm.tolist()) == list(range(n))
    gr = reorder.permute_graph(gs, perm)
    assert spread(gs) > 10 * spread(gr) and reorder.halo_fraction([gr], 8)[0] < 0.2 < reorder.halo_fraction([gs], 8)[0]
    wl = dict(x=torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 3), labels=torch.arange(n, dtype=torch.int32),
              train_mask=torch.ones(n, dtype=torch.uint8), val_mask=torch.zeros(n, dtype=torch.uint8), graphs=[gs])
    rel = reorder.relabel(wl)
    assert torch.equal(rel.wl["labels"].long(), rel.perm) and torch.equal(rel.unpermute(rel.wl["x"]), wl["x"])
    # two components + an isolated node
    rp = torch.tensor([0, 2, 4, 5, 7, 9], dtype=torch.int64)
    ci = torch.tensor([0, 1, 0, 1, 2, 3, 4, 3, 4], dtype=torch.int32)
    assert sorted(reorder.bfs_order([CSRGraph(rp, ci, 5)]).tolist()) == [0, 1, 2, 3, 4]
    # many small components and isolated nodes (real meta-path graphs): still a permutation, every component
    # contiguous, the nodes without a neighbour as one batch at the end in id order
    m = 400
    pairs = torch.randperm(m, generator=torch.Generator().manual_seed(5))[:300].reshape(150, 2)
    adj = torch.eye(m, dtype=torch.bool)
    adj[pairs[:, 0], pairs[:, 1]] = True
    adj[pairs[:, 1], pairs[:, 0]] = True
    gm = CSRGraph.from_bias(torch.where(adj, 0.0, -1e9)[None])
    pm = reorder.bfs_order([gm]).tolist()
    assert sorted(pm) == list(range(m))
    lonely = sorted(set(range(m)) - set(pairs.flatten().tolist()))
    assert pm[m - len(lonely):] == lonely
    pos = {v: i for i, v in enumerate(pm)}
    assert all(abs(pos[int(a)] - pos[int(b)]) == 1 for a, b in pairs)


def test_shard_graphs_measure_locality_against_their_own_rows():
    """CSRGraph.has_locality() on the shards of a node partition: rows are local (0..n_local), columns
    global -- the estimate subtracts the shard's first row (row_base), so every rank decides alike."""
    from han_amd import synth
    from han_amd.dist import NodePartition
    n = 4000
    g = synth.banded_graph(n, 8, 20, 3)
    assert g.has_locality()
    for r in range(4):
        part = NodePartition(n, r, 4)
        rl, cl = part.shard_graph(g)
        assert rl.row_base == part.row_start == cl.row_base
        assert rl.has_locality() and cl.has_locality(), r
    gr = synth.random_regular_graph(n, 10, 3)
    assert not any(x.has_locality() for x in NodePartition(n, 3, 4).shard_graph(gr))


def test_locality_pass_is_a_pure_relabelling_on_cpu_backend(cpu_ops):
    """Loss, accuracy and every parameter gradient are those of the original problem; per-node outputs
    come back in the original order (drop = 0: dropout masks are keyed by node id)."""
    from han_amd import ops, reorder
    prob = make_problem(17, 60, 7, 2, 3, [0.05, 0.2])
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    labels = torch.tensor(prob["labels"], dtype=torch.int32)
    mask = torch.tensor(prob["mask"].astype(np.uint8))
    graphs = _cpu_graphs(prob)
    res = []
    rel = reorder.relabel(dict(x=x, labels=labels, train_mask=mask, val_mask=mask, graphs=graphs))
    for wl, back in ((dict(x=x, labels=labels, train_mask=mask, graphs=graphs), lambda t: t), (rel.wl, rel.unpermute)):
        model, _ = _cpu_model(prob)
        model.zero_grad_flat()
        M = model.node_level([wl["x"]] * 2, wl["graphs"], 0.0, 0.0, True, ops.ACT_ELU)
        Z, att = model.semantic(M)
        loss, acc, logits = model.classifier_loss(Z, wl["labels"], wl["train_mask"], 1.0 / int(mask.sum()))
        loss.backward()
        res.append((float(loss.detach()), float(acc), back(logits.detach()), back(att.detach()), model.flat_grad.clone()))
    (l0, a0, lg0, at0, g0), (l1, a1, lg1, at1, g1) = res
    assert abs(l0 - l1) < 1e-6 and abs(a0 - a1) < 1e-7
    assert float((lg0 - lg1).abs().max()) < 1e-5 and float((at0 - at1).abs().max()) < 1e-6
    assert float((g0 - g1).abs().max()) < 1e-6 * max(1.0, float(g0.abs().max()))


def test_row_bins_partition_the_rows():
    """CSRGraph.row_bins (the degree bins of a K2 launch): every row in exactly one of short / mid / long, the short
    rows ordered by their number of 4-entry steps, a bin that holds every row is the identity (no list)."""
    from han_amd.graph import CSRGraph
    rng = np.random.default_rng(3)
    deg = rng.choice([0, 1, 3, 15, 16, 17, 40, 9000], size=400, p=[0.05, 0.2, 0.2, 0.15, 0.15, 0.1, 0.14, 0.01])
    deg[5] = 9000
    rp = np.zeros(401, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    g = CSRGraph(torch.tensor(rp), torch.zeros(int(rp[-1]), dtype=torch.int32), 400, validate=False)
    rb, sp = g.row_bins(16, 8192), g.row_split(8192, 4096)
    short, mid, long_ = rb["short_rows"].numpy(), rb["mid_rows"].numpy(), sp["long_rows"].numpy()
    assert sorted(np.concatenate([short, mid, long_]).tolist()) == list(range(400))
    assert (deg[short] < 16).all() and ((deg[mid] >= 16) & (deg[mid] <= 8192)).all() and (deg[long_] > 8192).all()
    assert rb["n_short"] == len(short) and rb["n_mid"] == len(mid)
    steps = (deg[short] + 3) // 4
    assert (np.diff(steps) >= 0).all()
    for s_ in np.unique(steps):                      # ids ascending inside a step class
        assert (np.diff(short[steps == s_]) > 0).all()
    assert (np.diff(mid) > 0).all()
    # one bin holds everything: identity, no list
    g2 = CSRGraph(torch.arange(0, 50 * 11, 50, dtype=torch.int64), torch.zeros(500, dtype=torch.int32), 10, validate=False)
    rb2 = g2.row_bins(16, 8192)
    assert rb2 == dict(n_short=0, short_rows=None, n_mid=10, mid_rows=None)
    g3 = CSRGraph(torch.arange(0, 3 * 11, 3, dtype=torch.int64), torch.zeros(30, dtype=torch.int32), 10, validate=False)
    assert g3.row_bins(16, 8192) == dict(n_short=10, short_rows=None, n_mid=0, mid_rows=None)


def test_inplace_change_of_the_node_level_output_raises(cpu_ops):
    """the K2 backward recovers the pre-activation from the forward's OUTPUT M; M is saved through
    ctx.save_for_backward, so an in-place op on it (or on a slice) between forward and backward is caught by autograd's
    version check instead of silently giving wrong gradients."""
    prob = make_problem(41, 30, 6, 2, 3, [0.2, 0.4])
    model, _ = _cpu_model(prob)
    graphs = _cpu_graphs(prob)
    x = torch.tensor(prob["x"][0], dtype=torch.float32)
    M = model.node_level([x, x], graphs, 0.0, 0.0, True, cpu_ops.ACT_ELU)
    M.sum().backward()                                   # untouched: fine
    M = model.node_level([x, x], graphs, 0.0, 0.0, True, cpu_ops.ACT_ELU)
    M[:, 0, :].mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        M.sum().backward()


def test_lds_bank_model_on_the_k3_backward_layouts():
    """tools/lds_banks.py restates the LDS banking rules of the MI355X guide (lane groups and bank width per
    instruction).  The round-3 K3 backward measured 46 % bank-conflict cycles; the model must say so for its three
    16-byte read patterns, and must find the round-4 layouts (sem_attn.hip: SA_WLDB = 160, rows of 288 B in G2's column
    order, dpre tile pitch 132 words with g2_col's k-slots) conflict-free, with the tile's 4-byte writes unchanged."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from lds_banks import cycles
    l15, l4 = (lambda l: l & 15), (lambda l: l >> 4)
    # round 3: rows of 144 B / 272 B, dpre pieces at 32 s + 8 l4 floats
    assert cycles(lambda l: l15(l) * 144 + 16 * l4(l), "read_b128")[0] == 8
    assert cycles(lambda l: l15(l) * 272 + 16 * l4(l), "read_b128")[0] == 8
    assert cycles(lambda l: (l15(l) * 132 + 8 * l4(l)) * 4, "read_b128")[0] == 8
    # round 4
    g2_col = lambda s, q: 64 * (q & 1) + 32 * (q >> 1) + 8 * s
    for s2 in range(4):
        assert cycles(lambda l: l15(l) * 160 + 64 * (s2 & 1) + 16 * l4(l), "read_b128")[0] == 4       # G1: Womega^T fragments
        assert cycles(lambda l: l15(l) * 288 + (4 * s2 + l4(l)) * 16, "read_b128")[0] == 4            # G2: Womega fragments
        for half in (0, 4):
            assert cycles(lambda l: (l15(l) * 132 + g2_col(s2, l4(l)) + half) * 4, "read_b128")[0] == 4   # dpre tile
    for t in range(8):
        for reg in range(4):
            assert cycles(lambda l: ((4 * l4(l) + reg) * 132 + 16 * t + l15(l)) * 4, "write_b32")[0] == 2
    # a pitch that is a multiple of 256 B serialises a 16-lane group completely
    assert cycles(lambda l: l15(l) * 256 + 16 * l4(l), "read_b128")[0] == 32

