"""world_size-2 gloo run of the node-partitioned trainer on CPU (kernels replaced
by tests/cpu_backend.py): the partitioned epochs must reproduce the
single-process epochs -- identical dropout masks through global-id RNG keys,
all-gathered tables, one all-reduce of the flat gradient buffer."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, epochs, drop, out, port, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r),
                               str(world), "cpu", str(epochs), str(drop), out, str(port), "1"],
                              env=env, cwd=ROOT) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0


@pytest.mark.parametrize("drop,port", [(0.0, 29701), (0.6, 29711)])
def test_two_gloo_ranks_match_single_process(tmp_path, drop, port):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _launch(1, 2, drop, one, port)
    _launch(2, 2, drop, two, port + 2)
    a, b = np.load(one), np.load(two)
    assert np.isfinite(a["flat"]).all()
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


@pytest.mark.parametrize("weighted,port", [("0", 29721), ("1", 29731)])
def test_halo_exchange_on_a_graph_with_locality(tmp_path, weighted, port):
    """Banded graphs: only a few boundary rows are remote, so the trainer plans a halo
    exchange (send lists + all-to-all-v + remapped colidx + global-id RNG keys) instead
    of the all-gather; results must still equal the single-process run.  weighted: the
    graphs carry sp_attn_head edge values, which must follow the rows into the shards,
    the transposed shards and the remapped halo graphs."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_GRAPH": "band", "HAN_TEST_WEIGHTED": weighted}
    _launch(1, 2, 0.6, one, port, env)
    _launch(2, 2, 0.6, two, port + 2, env)
    a, b = np.load(one), np.load(two)
    assert int(b["halo_plans"]) == 4          # 2 meta-paths x (forward, backward) all in halo mode
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


def test_four_gloo_ranks_uneven_shards(tmp_path):
    """world_size 4 with N = 257 (shards of 65, 65, 65, 62 rows): the driver also runs 4 and 8
    ranks.  Halo mode on the banded graphs; must equal the single-process run."""
    one, four = str(tmp_path / "one.npz"), str(tmp_path / "four.npz")
    env = {"HAN_TEST_GRAPH": "band"}
    _launch(1, 2, 0.6, one, 29741, env)
    _launch(4, 2, 0.6, four, 29743, env)
    a, b = np.load(one), np.load(four)
    assert int(b["halo_plans"]) == 4
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


def test_eight_gloo_ranks_mixed_exchange_modes(tmp_path):
    """world_size 8 (the driver's largest run; shards of 33 ... 26 rows): the sparse meta-path
    plans a halo exchange, the dense one falls back to the all-gather; must equal the
    single-process run."""
    one, eight = str(tmp_path / "one.npz"), str(tmp_path / "eight.npz")
    _launch(1, 2, 0.6, one, 29761, {"OMP_NUM_THREADS": "1"})
    _launch(8, 2, 0.6, eight, 29763, {"OMP_NUM_THREADS": "1"})
    a, b = np.load(one), np.load(eight)
    assert 0 < int(b["halo_plans"]) < 4
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


@pytest.mark.parametrize("bf16,port", [("0", 29781), ("1", 29791)])
def test_eight_gloo_ranks_all_gather_everywhere_from_local_rows(tmp_path, bf16, port):
    """world_size 8, N = 257 (uneven shards), every meta-path forced onto the ALL-GATHER path (what the
    uniformly random SYN-1M / SYN-10M graphs use), every rank holding only its OWN graph rows
    (NodePartition.shard_local_graph: transposed shards from an all-to-all-v of the edges) and -- bf16 --
    the configs[4] storage (bf16 X / H tables, fused bf16 [g | stats] rows on the wire).  Must equal the
    single-process run on the global graph."""
    one, eight = str(tmp_path / "one.npz"), str(tmp_path / "eight.npz")
    env = {"OMP_NUM_THREADS": "1", "HAN_TEST_ALLGATHER": "1", "HAN_TEST_LOCAL": "1", "HAN_TEST_BF16": bf16}
    _launch(1, 2, 0.6, one, port, env)
    _launch(8, 2, 0.6, eight, port + 2, env)
    a, b = np.load(one), np.load(eight)
    assert int(b["halo_plans"]) == 0
    tol = 1e-5 if bf16 == "0" else 2e-4      # bf16: a sum-order difference can flip one stored rounding
    assert np.abs(a["flat"] - b["flat"]).max() < tol
    assert np.abs(a["hist"] - b["hist"]).max() < tol


def test_two_gloo_ranks_local_rows_halo_weighted(tmp_path):
    """Rank-local rows + halo plans + sp_attn_head edge values: the values must travel with the edges
    through the all-to-all-v that builds the transposed shards."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_GRAPH": "band", "HAN_TEST_WEIGHTED": "1", "HAN_TEST_LOCAL": "1"}
    _launch(1, 2, 0.6, one, 29801, env)
    _launch(2, 2, 0.6, two, 29803, env)
    a, b = np.load(one), np.load(two)
    assert int(b["halo_plans"]) == 4
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


@pytest.mark.parametrize("world,mode,bf16,port", [(2, "all", "0", 29821), (4, "auto", "0", 29831),
                                                    (8, "auto", "1", 29841), (8, "eval", "0", 29851)])
def test_replicated_projection_matches_single_process(tmp_path, world, mode, bf16, port):
    """Recompute instead of communicate: every rank holds the features of all N = 257 rows (uneven
    shards -> the full table is padded to world * shard rows) and projects the whole H table itself in
    the forward passes the policy names (both up to 4 ranks, the eval forward only at 8), so only the
    backward [g | stats] table and the gradients travel.  Rank-local graph rows, all-gather path.
    Dropout masks are keyed by global row ids, hence the same epochs as a single process."""
    one, many = str(tmp_path / "one.npz"), str(tmp_path / "many.npz")
    env = {"OMP_NUM_THREADS": "1", "HAN_TEST_ALLGATHER": "1", "HAN_TEST_LOCAL": "1", "HAN_TEST_BF16": bf16}
    _launch(1, 2, 0.6, one, port, env)
    _launch(world, 2, 0.6, many, port + 2, dict(env, HAN_TEST_REPLICATE=mode))
    a, b = np.load(one), np.load(many)
    assert str(b["replicate"]) == ("eval,train" if (mode == "all" or (mode == "auto" and world <= 4)) else "eval")
    tol = 1e-5 if bf16 == "0" else 2e-4
    assert np.abs(a["flat"] - b["flat"]).max() < tol
    assert np.abs(a["hist"] - b["hist"]).max() < tol


def test_replicated_projection_keeps_halo_plans(tmp_path):
    """A meta-path with locality keeps its halo exchange (a few boundary rows are cheaper to receive than
    a whole table is to project); the others are projected in full."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"OMP_NUM_THREADS": "1"}
    _launch(1, 2, 0.6, one, 29861, env)
    _launch(8, 2, 0.6, two, 29863, dict(env, HAN_TEST_REPLICATE="all"))
    a, b = np.load(one), np.load(two)
    assert 0 < int(b["halo_plans"]) < 4
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


@pytest.mark.parametrize("weighted", ["0", "1"])
def test_masked_backward_is_bit_identical_to_the_full_pass(tmp_path, weighted):
    """Opt-in masked backward (HANTrainer(masked_backward=True)): with one node-attention layer the destinations
    outside the train mask have g == 0, the transposed-graph pass skips them IN PLACE and only the live rows of
    the backward table [g | stats] travel (one all-gather of world x max-live rows with a global-id table for
    the dropout keys).  Parameters after 3 epochs with dropout: bit-equal between the masked and the full pass
    at world 8 (uneven shards: 33 rows x 7 + 26) and at world 1 -- and world 8 equal to world 1 as before."""
    outs = {}
    for tag, world, masked, port in (("full1", 1, "0", 29851), ("mask1", 1, "1", 29853),
                                     ("full8", 8, "0", 29855), ("mask8", 8, "1", 29865)):
        out = str(tmp_path / f"{tag}.npz")
        _launch(world, 3, 0.6, out, port + (10 if weighted == "1" else 0) * 4,
                {"HAN_TEST_MASKED_BWD": masked, "HAN_TEST_ALLGATHER": "1", "HAN_TEST_WEIGHTED": weighted})
        outs[tag] = np.load(out)
    assert np.isfinite(outs["full1"]["flat"]).all()
    assert np.array_equal(outs["full1"]["flat"], outs["mask1"]["flat"])
    assert np.array_equal(outs["full8"]["flat"], outs["mask8"]["flat"])
    assert np.array_equal(outs["full8"]["hist"], outs["mask8"]["hist"])
    assert np.abs(outs["full1"]["flat"] - outs["full8"]["flat"]).max() < 1e-5


@pytest.mark.parametrize("graph,port", [("", 29871), ("band", 29881)])
def test_wide_heads_under_a_node_partition(tmp_path, graph, port):
    """hid_units = [96] (two heads of two 64-column slices each, layers.WideHeadAttention) on 2 ranks: every slice's
    table and the heads' f2 totals are exchanged (all-gather on the random graphs, halo plans on the banded ones), the
    backward moves one [g | stats] table per slice; parameters and metrics equal the single-process run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_HID": "96"}
    if graph:
        env["HAN_TEST_GRAPH"] = graph
    _launch(1, 2, 0.6, one, port, env)
    _launch(2, 2, 0.6, two, port + 2, env)
    a, b = np.load(one), np.load(two)
    assert np.isfinite(a["flat"]).all()
    if graph:
        assert int(b["halo_plans"]) == 4
    assert np.abs(a["flat"] - b["flat"]).max() < 1e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-5


@pytest.mark.parametrize("mode,port", [("allgather", 29791), ("halo", 29795)])
def test_comm_stats_count_what_a_rank_receives(tmp_path, mode, port):
    """dist.CommStats (bench.py's N > 1 keys exchange_bytes_received / comm_wait_ms / grad_allreduce_bytes): per epoch and
    meta-path a rank receives one forward table in the training step, one in the eval forward (256 B per row) and one
    fused [g | stats] table in the backward (384 B per row) -- the other ranks' whole blocks in all-gather mode, its halo
    rows in halo mode -- and all-reduces the flat gradient buffer once (reduce_metrics adds 16 B per epoch)."""
    out = str(tmp_path / "two.npz")
    env = {"HAN_TEST_COMM": "1"}
    env.update({"HAN_TEST_ALLGATHER": "1"} if mode == "allgather" else {"HAN_TEST_GRAPH": "band"})
    epochs, world, p = 2, 2, 2
    _launch(world, epochs, 0.6, out, port, env)
    z = np.load(out)
    assert int(z["comm_exchanges"]) == epochs * p * 3
    if mode == "allgather":
        assert int(z["halo_plans"]) == 0
        assert int(z["comm_bytes"]) == epochs * p * (world - 1) * int(z["shard"]) * (256 + 256 + 384)
    else:
        assert int(z["halo_plans"]) == 4
        # forward plans serve two exchanges per epoch (train + eval), backward plans one; halo_rows sums both kinds
        assert 0 < int(z["comm_bytes"]) < epochs * p * int(z["shard"]) * (256 + 256 + 384) // 4
    assert int(z["comm_allreduce"]) == epochs * (int(z["n_params"]) * 4 + 4 * 4)
    assert float(z["comm_wait_ms"]) > 0.0       # host-staged gloo collectives: wall time on the host
