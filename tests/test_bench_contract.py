"""bench.py's JSON line: the helpers that price the K2 kernels, and the committed lines under
profiles/ against the driver's contract (keys, units, no HBM fraction above 1, a bounded
CPU baseline with both thread counts).  No GPU needed."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _lines(rnd="r02"):
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_bench_*.json"))):
        txt = [l for l in open(f).read().splitlines() if l.startswith("{")]
        assert len(txt) == 1, f"{f}: expected exactly one JSON line"
        out.append((os.path.basename(f), json.loads(txt[0])))
    return out


def test_k2_byte_model():
    # eval forward = E*(4 + 256 + 32) + rows; what the kernels request drops the f2 gather
    n, e = 1_000_000, 50_000_000
    b = bench.k2_bytes("eval", n, e, s=4)
    assert b["algorithmic"] - b["moved"] == e * 32
    assert b["moved"] >= e * (4 + 256)
    assert b["compulsory"] < b["moved"]
    bw = bench.k2_bytes("bwd_cols", n, e, s=4)
    assert bw["moved"] >= e * (4 + 384)            # index + the fused [g | stats] row
    b16 = bench.k2_bytes("eval", n, e, s=2)
    assert b16["moved"] < b["moved"]


def test_static_traffic_is_tied_to_the_kernel_sources(monkeypatch):
    per, src = bench.static_traffic("syn-1m", 2, "f32")      # measured on one GPU only
    assert per is None
    per, src = bench.static_traffic("syn-1m", 1, "bf16")
    assert per is None
    monkeypatch.setattr(bench, "_src_sha", lambda: "0" * 16)
    per, why = bench.static_traffic("syn-1m", 1, "f32")
    assert per is None and "different kernel sources" in why


def test_committed_bench_lines_follow_the_contract():
    lines = _lines()
    assert any(name == "r02_bench_syn1m_f32_1gpu.json" for name, _ in lines)
    for name, d in lines:
        d.setdefault("roofline", None)    # hipGraph-mode lines committed before the key became unconditional
        assert CONTRACT_KEYS <= set(d), (name, CONTRACT_KEYS - set(d))
        assert d["unit"] == "epochs/s" and d["higher_is_better"] is True and d["data"] == "synthetic"
        assert d["vs_baseline"] is None                       # BASELINE.md holds no published number
        assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        if r is None:                                         # a replayed hipGraph records no per-kernel events
            assert "hipGraph" in d["config"]["parallelism"]
            continue
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r), name
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        if r["bound"] == "hbm":
            assert r["frac"] <= 1.0, (name, r["frac"])
        for v in (d.get("roofline_hbm_regime") or {}).values():
            if isinstance(v, dict):
                assert v["frac"] <= 1.0


def test_headline_line_carries_traffic_and_both_cpu_baselines():
    d = dict(_lines())["r02_bench_syn1m_f32_1gpu.json"]
    assert d["n_gpus"] == 1 and d["dtype"].startswith("f32")
    r = d["roofline"]
    assert r["traffic"] and r["traffic_source"]      # live PMC passes, or the sha-matched committed figure
    # measured fabric bytes within 2 % of the bytes the kernel requests
    assert abs(r["traffic"] / r["moved_bytes_per_launch"] - 1.0) < 0.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "epochs/s"
    assert c["all_cores"]["threads"] == c["cores"] and c["one_thread"]["threads"] == 1
    assert c["all_cores"]["sample_n"] >= 50_000
    hb = d["roofline_hbm_regime"]
    assert all(hb[k]["frac"] >= 0.5 for k in ("eval", "train", "bwd_cols"))     # BASELINE: >= 50 % of HBM peak


def test_host_cores_reports_usable_cores():
    cores, affinity, quota = bench.host_cores()
    assert 1 <= cores <= affinity <= (os.cpu_count() or 1)
    assert quota is None or cores <= quota


def test_round3_headline_line_is_complete_and_self_consistent():
    """the default line carries the power-law variant beside the headline, a cache-path
    K2 rate is never labelled an HBM fraction (bound 'fabric / infinity cache' + the HBM-regime fraction of the same
    kernel in hbm_frac), the PMC traffic belongs to the headline workload only, and the CPU baseline says what the
    cores really bought (one thread and all cores on the SAME sample)."""
    lines = dict(_lines("r03"))
    d = lines["r03_bench_syn1m_f32_1gpu.json"]
    assert CONTRACT_KEYS <= set(d) and d["unit"] == "epochs/s" and d["vs_baseline"] is None
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3
    r = d["roofline"]
    assert r["bound"] == "fabric / infinity cache" and 0.5 <= r["hbm_frac"] <= 1.0       # BASELINE: >= 50 % of the HBM roofline
    assert abs(r["hbm_frac"] - d["roofline_hbm_regime"][[k for k, v in d["roofline_k2_all"].items()
                                                         if v["kernel"] == r["kernel"]][0]]["frac"]) < 1e-9
    assert abs(r["traffic"] / r["moved_bytes_per_launch"] - 1.0) < 0.03
    for v in d["roofline_k2_all"].values():
        assert v["bound"] != "hbm" and v["hbm_frac"] >= 0.5
    sk = d["skew"]
    assert sk["unit"] == "epochs/s" and "power-law" in sk["workload"] and set(sk["k2"]) == {"eval", "train", "bwd_cols"}
    assert abs(sk["value"] * sk["ms_per_step"] / 1e3 - 1.0) < 1e-3
    assert all(v["bound"] != "hbm" for v in sk["k2"].values())
    c = d["cpu_baseline"]
    assert c["all_cores"]["sample_n"] == c["one_thread"]["sample_n"] >= 10_000
    assert c["all_cores"]["threads"] == c["cores"] and c["one_thread"]["threads"] == 1
    assert abs(c["cores_effective"] - c["one_thread"]["epoch_s"] / c["all_cores"]["epoch_s"]) < 0.02
    assert d["value"] >= 26.3                         # not slower than round 2's line
    for name, dd in lines.items():                    # every committed round-3 line follows the contract
        assert CONTRACT_KEYS <= set(dd) | {"roofline"}, name
        assert dd["unit"] == "epochs/s" and dd["data"] == "synthetic" and "workload" in dd["config"]


def test_cpu_baseline_leg_runs_on_one_sample(monkeypatch):
    """bench.cpu_baseline on a tiny sample: both thread counts on the same N, cores_effective = their ratio."""
    monkeypatch.setattr(bench, "host_cores", lambda: (2, 2, None))
    monkeypatch.setattr(bench, "_tune_host_malloc", lambda: False)      # do not retune the test process's allocator
    c = bench.cpu_baseline("tiny", 512, 512)
    assert c["all_cores"]["sample_n"] == c["one_thread"]["sample_n"] == 512
    assert c["cores"] == 2 and c["one_thread"]["threads"] == 1 and c["cores_effective"] > 0
    assert c["kind"] == "port" and c["value"] > 0


def test_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus N` (no torchrun, no RANK in the environment) must start N ranks by
    itself -- the parent passes its arguments through unchanged, relays the children's exit code, and is a no-op when
    it IS a rank or N = 1."""
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--nodes", "200000"]
    calls = []

    def fake_run(cmd, env=None, cwd=None):
        calls.append((cmd, env, cwd))
        return 7

    assert bench.launch_ranks_if_needed(argv, environ={"PATH": "/usr/bin"}, run=fake_run) == 7      # exit code relayed
    (cmd, env, cwd), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv       # pass-through
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and cwd == ROOT
    # already a rank (torchrun set RANK), or one GPU: nothing to launch
    assert bench.launch_ranks_if_needed(argv, environ={"RANK": "0", "WORLD_SIZE": "4"}, run=fake_run) is None
    assert bench.launch_ranks_if_needed(["--steps", "2"], environ={}, run=fake_run) is None
    assert bench.launch_ranks_if_needed(["--gpus=1"], environ={}, run=fake_run) is None
    assert bench._gpus_from_argv(["--gpus=8", "--steps", "1"]) == 8
    assert len(calls) == 1


def test_launcher_relays_a_failing_child():
    """The real launcher path, end to end without a GPU: a child that dies (here: --gpus 2 with an invalid workload
    makes every rank exit non-zero before it touches a device) must make the parent exit non-zero."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "no-such-workload",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no JSON line from a failed job


MULTI_RANK_KEYS = {"rccl_ranks", "backend", "exchange_bytes_received", "comm_wait_ms", "grad_allreduce_bytes"}


def test_multi_rank_lines_carry_the_exchange_keys():
    """Every committed N > 1 line of round 4 says how many ranks really ran, over what backend, how many bytes each
    received per step and how long the compute stream waited for them."""
    seen = 0
    for name, d in _lines("r04"):
        if d["n_gpus"] > 1:
            seen += 1
            assert MULTI_RANK_KEYS <= set(d), (name, MULTI_RANK_KEYS - set(d))
            assert d["rccl_ranks"] == d["n_gpus"] and d["exchange_bytes_received"] > 0 and d["comm_wait_ms"] >= 0
    assert seen >= 1


def test_round4_lines_meet_what_they_claim():
    """The committed round-4 lines: the default line with the power-law variant's K2 launches at the targets VERDICT r3
    set (training forward <= 2.0 ms, eval <= 1.85 ms) and the uniform headline not slower than round 3; the small-graph
    lines on graphs with EXACTLY the data sets' entry counts; the DBLP-like line says which meta-path ran K2 in the dense
    form and beats the CSR-only run of the same build."""
    lines = dict(_lines("r04"))
    d = lines["r04_bench_syn1m_f32_1gpu.json"]
    assert CONTRACT_KEYS <= set(d) and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3 and d["value"] >= 27.5
    r = d["roofline"]
    assert r["bound"] == "fabric / infinity cache" and 0.5 <= r["hbm_frac"] <= 1.0
    assert abs(r["traffic"] / r["moved_bytes_per_launch"] - 1.0) < 0.03
    assert all(v["frac"] >= 0.5 for v in d["roofline_hbm_regime"].values())          # BASELINE: >= 50 % of the HBM roofline
    sk = d["skew"]["k2"]
    assert sk["train"]["avg_launch_ms"] <= 2.0 and sk["eval"]["avg_launch_ms"] <= 1.85
    assert d["skew"]["value"] >= 30.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["all_cores"]["sample_n"] == c["one_thread"]["sample_n"] >= 10_000
    dblp, dblp_csr, acm = (lines[k] for k in ("r04_bench_dblp_like_graph.json", "r04_bench_dblp_like_graph_no_dense.json",
                                              "r04_bench_acm_like_graph.json"))
    assert "E=17936007 [11113, 5000495, 12924399]" in dblp["config"]["workload"]
    assert "E=2240042 [29281, 2210761]" in acm["config"]["workload"]
    assert dblp["config"]["k2_dense_form"] == [False, False, True] and dblp_csr["config"]["k2_dense_form"] == [False] * 3
    assert dblp["ms_per_step"] <= 1.0 and dblp["value"] > 1.15 * dblp_csr["value"]
    assert acm["config"]["k2_dense_form"] == [False, False]
    # overlap_eval: the same epochs (bit-equal training pair after 320 of them), faster than the plain epoch of the same box
    for w in ("acm", "dblp"):
        ov, pl = lines[f"r04_bench_{w}_like_graph_overlap_eval.json"], lines[f"r04_bench_{w}_like_graph_plain_same_box.json"]
        assert "overlap_eval" in ov["config"]["parallelism"] and "overlap_eval" not in pl["config"]["parallelism"]
        assert ov["config"]["workload"] == pl["config"]["workload"] and ov["steps"] == pl["steps"]
        assert ov["final"]["train_loss"] == pl["final"]["train_loss"] and ov["final"]["train_acc"] == pl["final"]["train_acc"]
        assert ov["value"] > (1.15 if w == "acm" else 1.03) * pl["value"]
    b16 = lines["r04_bench_syn10m_bf16_1gpu.json"]
    assert "syn-10m" in b16["config"]["workload"] and b16["dtype"].startswith("bf16") and b16["value"] > 1.5
    for name, dd in lines.items():
        assert CONTRACT_KEYS <= set(dd) | {"roofline"}, name
        assert dd["unit"] == "epochs/s" and dd["data"] == "synthetic" and "workload" in dd["config"]
