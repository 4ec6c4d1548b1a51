"""Node-partitioned training on REAL kernels with 2 ranks sharing the one GPU
(gloo collectives staged through the host): the partitioned run must reproduce
the single-process run -- same dropout masks (global-id RNG keys), same
parameters after several epochs."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, epochs, drop, out, port, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r),
                               str(world), "cuda:0", str(epochs), str(drop), out, str(port), "0"],
                              env=env, cwd=ROOT) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0


@pytest.mark.parametrize("drop", [0.0, 0.6])
def test_two_ranks_match_single_process(dev, tmp_path, drop):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _launch(1, 3, drop, one, 29611)
    _launch(2, 3, drop, two, 29613)
    a, b = np.load(one), np.load(two)
    assert np.isfinite(a["flat"]).all()
    # identical algorithm, different summation order in the cross-rank reductions
    assert np.abs(a["flat"] - b["flat"]).max() < 2e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 2e-5


@pytest.mark.parametrize("mode", ["halo", "allgather-local-rows", "allgather-local-rows-bf16",
                                  "allgather-local-rows-replicate-auto"])
def test_rccl_collectives_on_a_one_rank_group(dev, tmp_path, mode):
    """The RCCL code path on a 1-rank communicator -- the only way to execute those calls on a single-GPU
    box: all_to_all_single with (empty) uneven splits for the halo plans and their per-step exchange,
    all_gather_into_tensor async + wait into the persistent tables (fp32 H, fused uint8 [g | stats] rows,
    bf16 storage), the edge all-to-all-v of shard_local_graph, the all_reduce of the flat gradients, barrier."""
    one, forced = str(tmp_path / "one.npz"), str(tmp_path / "forced.npz")
    extra = {}
    if mode != "halo":
        extra = {"HAN_TEST_ALLGATHER": "1", "HAN_TEST_LOCAL": "1",
                 "HAN_TEST_BF16": "1" if mode.endswith("bf16") else "0"}
    forced_extra = {}
    if mode.endswith("replicate-auto"):
        # replicate="auto" over RCCL measures exchange against recompute at set-up (HANTrainer._calibrate_replication)
        forced_extra = {"HAN_TEST_REPLICATE": "auto"}
    _launch(1, 2, 0.6, one, 29621, extra)
    _launch(1, 2, 0.6, forced, 29623, {"HAN_FORCE_COLLECTIVES": "1", "HAN_TEST_BACKEND": "nccl", **extra,
                                       **forced_extra})
    a, b = np.load(one), np.load(forced)
    tol = 5e-6 if not mode.endswith("bf16") else 5e-4      # different kernel instantiations (FAST / generic)
    assert np.abs(a["flat"] - b["flat"]).max() < tol
    assert np.abs(a["hist"] - b["hist"]).max() < max(tol, 1e-6)
    assert int(b["halo_plans"]) == (4 if mode == "halo" else 0)
    if mode.endswith("replicate-auto"):
        assert str(b["replicate"]) == "eval,train"      # one rank: projecting "all" rows costs nothing extra


def test_halo_exchange_two_ranks_real_kernels(dev, tmp_path):
    """Halo mode (banded graphs) on the real kernels: [local | halo] tables, remapped
    colidx, table_gid RNG keys, with dropout; equals the single-process run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_GRAPH": "band"}
    _launch(1, 3, 0.6, one, 29631, env)
    _launch(2, 3, 0.6, two, 29633, env)
    a, b = np.load(one), np.load(two)
    assert int(b["halo_plans"]) == 4
    assert np.abs(a["flat"] - b["flat"]).max() < 2e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 2e-5


@pytest.mark.parametrize("bf16", ["0", "1"])
def test_local_rows_all_gather_two_ranks_real_kernels(dev, tmp_path, bf16):
    """Every rank holds only its own graph rows (transposed shards from the edge all-to-all-v), every
    meta-path on the all-gather path, fp32 and the configs[4] bf16 storage (fused bf16 [g | stats] rows
    and bf16 H tables on the wire), with dropout, on the real kernels; equals the single-process run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_ALLGATHER": "1", "HAN_TEST_LOCAL": "1", "HAN_TEST_BF16": bf16}
    _launch(1, 3, 0.6, one, 29641, env)
    _launch(2, 3, 0.6, two, 29643, env)
    a, b = np.load(one), np.load(two)
    assert int(b["halo_plans"]) == 0
    tol = 2e-5 if bf16 == "0" else 5e-4
    assert np.abs(a["flat"] - b["flat"]).max() < tol
    assert np.abs(a["hist"] - b["hist"]).max() < tol


@pytest.mark.parametrize("mode,bf16", [("all", "0"), ("eval", "0"), ("all", "1")])
def test_replicated_projection_two_ranks_real_kernels(dev, tmp_path, mode, bf16):
    """Recompute instead of communicate, on the real kernels: both ranks hold the features of all
    257 rows (uneven shards: the table is padded to world * shard rows) and project the whole H table
    themselves in the forward passes named by `mode`; K2 reads that table, the backward works on the
    local row slice of it; only [g | stats] and the gradients travel.  Equals the single-process run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_ALLGATHER": "1", "HAN_TEST_LOCAL": "1", "HAN_TEST_BF16": bf16}
    _launch(1, 3, 0.6, one, 29651, env)
    _launch(2, 3, 0.6, two, 29653, dict(env, HAN_TEST_REPLICATE=mode))
    a, b = np.load(one), np.load(two)
    assert str(b["replicate"]) == ("eval,train" if mode == "all" else "eval")
    tol = 2e-5 if bf16 == "0" else 5e-4
    assert np.abs(a["flat"] - b["flat"]).max() < tol
    assert np.abs(a["hist"] - b["hist"]).max() < tol


@pytest.mark.parametrize("bf16", ["0", "1"])
def test_masked_backward_real_kernels_bit_identical(dev, tmp_path, bf16):
    """Opt-in masked backward on the real kernels (HAN_FLAG_MASKED_EDGES: dead entries of the transposed graph skipped
    in place): one process and two ranks sharing the GPU (compact all-gather of the live [g | stats] rows with
    a global-id table for the dropout keys), with dropout, fp32 and bf16 tables.  Parameters after 3 epochs
    within the usual tolerance of the single-process full pass (whose backward runs the FAST instantiation;
    bit-equality inside one instantiation is pinned by test_masked_edges_backward_is_bit_identical and, for the
    host logic at world 8, by tests/test_dist_gloo.py)."""
    env = {"HAN_TEST_ALLGATHER": "1", "HAN_TEST_LOCAL": "1", "HAN_TEST_BF16": bf16}
    outs = {}
    for tag, world, masked, port in (("full1", 1, "0", 29661), ("mask1", 1, "1", 29663),
                                     ("full2", 2, "0", 29665), ("mask2", 2, "1", 29667)):
        out = str(tmp_path / f"{tag}.npz")
        _launch(world, 3, 0.6, out, port, dict(env, HAN_TEST_MASKED_BWD=masked))
        outs[tag] = np.load(out)
    tol = 2e-5 if bf16 == "0" else 5e-4
    assert np.isfinite(outs["mask1"]["flat"]).all()
    assert np.abs(outs["full1"]["flat"] - outs["mask1"]["flat"]).max() < tol
    assert np.abs(outs["full1"]["flat"] - outs["mask2"]["flat"]).max() < tol
    assert np.abs(outs["full1"]["hist"] - outs["mask2"]["hist"]).max() < tol


@pytest.mark.parametrize("graph", ["", "band"])
def test_wide_heads_two_ranks_real_kernels(dev, tmp_path, graph):
    """hid_units = [96] under a node partition on the real kernels (layers.WideHeadAttention): per slice one exchanged
    H table and one [g | stats] table, the heads' f2 totals gathered by K2 from their own exchanged table; all-gather
    and halo mode, with dropout; equals the single-process run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = {"HAN_TEST_HID": "96"}
    if graph:
        env["HAN_TEST_GRAPH"] = graph
    _launch(1, 3, 0.6, one, 29691, env)
    _launch(2, 3, 0.6, two, 29693, env)
    a, b = np.load(one), np.load(two)
    assert np.isfinite(a["flat"]).all()
    assert np.abs(a["flat"] - b["flat"]).max() < 2e-5
    assert np.abs(a["hist"] - b["hist"]).max() < 2e-5


def test_bench_gpus_2_starts_two_ranks_by_itself(dev):
    """`python bench.py --gpus 2 ...` with no torchrun in the command and no RANK in the
    environment starts its two ranks itself (here both on the one GPU, collectives staged through the host)
    and rank 0's line says how many ranks ran and what they exchanged."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HAN_DIST_BACKEND="gloo", HAN_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nodes", "200000",
                        "--steps", "3", "--warmup", "1"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["exchange_bytes_received"] > 0 and d["comm_wait_ms"] >= 0 and d["grad_allreduce_bytes"] > 0
    assert "node-partition x2" in d["config"]["parallelism"]
    assert np.isfinite(d["final"]["train_loss"])


def test_bench_default_path_prints_one_complete_line(dev):
    """The line the driver records: `python bench.py` on one GPU, every leg of the default path switched on (live
    K2 rooflines, the HBM-regime probe that releases the trainer first, the CPU baseline leg; the skew variant only
    runs at the full size) at a small size -- one JSON line with the contract's keys, the roofline and cpu_baseline objects."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nodes", "60000", "--steps", "2", "--warmup", "1",
                        "--hbm-regime-nodes", "100000", "--cpu-sample", "2000", "--traffic", "static"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    # ("hbm", or the label of a table that the Infinity Cache / the L2s serve: the same 8 TB/s peak is the yardstick)
    assert d["roofline"]["bound"] in ("hbm", "fabric / infinity cache", "l2 / infinity cache")
    assert d["roofline"]["peak"] == 8000.0 and d["roofline"]["frac"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "roofline_hbm_regime" in d
