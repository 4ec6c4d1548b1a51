#!/usr/bin/env python3
"""Headline benchmark: full-graph HAN train epochs/s + node-attention (K2) bandwidth.

    python bench.py --gpus N --steps K --warmup W [--workload syn-1m]

One "step" is one reference epoch (ex_acm3025.py:171-218): one full-graph
fwd+bwd+Adam step with dropout 0.6/0.6 plus one eval forward.  The workload is
BASELINE.json's configs[3] (SYN-1M: 1M nodes, 4 meta-paths, deg 50, 256-d feats,
8 heads x 8), synthetic data, random-init weights, node-partitioned over N GPUs
(strong scaling: the graph is fixed, each rank owns N/G rows).

Rank 0 prints ONE JSON line.  `roofline` prices the K2 (node-attention) kernel with the
largest total time in the timed region from the bytes it requests per launch (`moved`:
E*(4+256) + rows; SURVEY.md section 8d's 292 B/edge figure, which still counts the f2 gather
this implementation no longer performs, is given beside it) over its mean launch duration,
measured with HIP events on the launch stream inside the timed region; `roofline_k2_all`
has all three K2 kernels.  At SYN-1M the 256 MB gather table sits in the 256 MiB Infinity
Cache, so those rates are cache-path rates; `roofline_hbm_regime` re-times the same three
kernels in the same run on one meta-path of N = 10M rows (2.56 GB table): THAT is the HBM
fraction.  `cpu_baseline` is the torch-CPU port of the reference algorithm
(oracle/han_oracle_torch.py, CSR form) timed on this host on a bounded sample, with all
usable cores and with one thread on the same sample (`cores_effective` = the measured ratio).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# --------------------------------------------------------------------------- rank launcher
def _gpus_from_argv(argv):
    """--gpus N / --gpus=N from the raw argument list (stdlib only: this runs before torch is imported)."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(argv, port=None):
    """The command that starts the N ranks of `python bench.py --gpus N ...`: torch.distributed.run on this
    very file with the caller's arguments passed through unchanged (one process per GPU, rendezvous on 127.0.0.1)."""
    n = _gpus_from_argv(argv)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.abspath(__file__)] + list(argv)


def launch_ranks_if_needed(argv, environ=None, run=None):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment (i.e. NOT already started by
    torchrun): this process becomes a pure parent -- it starts the N ranks as child processes, lets rank 0's
    JSON line through on the inherited stdout, and returns the children's exit code.  It runs before torch is
    imported and never touches the GPU; nothing is exec'd.  Returns None when there is nothing to launch
    (N = 1, or this process IS a rank)."""
    environ = os.environ if environ is None else environ
    n = _gpus_from_argv(argv)
    if n <= 1 or "RANK" in environ:
        return None
    import subprocess
    env = dict(environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(argv)
    rc = (run or subprocess.call)(cmd, env=env, cwd=ROOT)
    return int(rc)


if __name__ == "__main__":
    _rc = launch_ranks_if_needed(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)
HBM_COPY_GBS = 6290.0      # same guide: what a float4 streaming copy reaches; random whole-row gathers: 5.5-5.8 TB/s


# --------------------------------------------------------------------------- K2 byte models
def k2_bytes(tag, n_rows, nnz, s=4, K=8, D=64, stats_row=None):
    """Bytes per launch of the three K2 kernels (s = bytes per gathered table element).

    `algorithmic` is SURVEY.md 8d's figure: per edge colidx 4 + neighbour row D*s + f2_j 4K.
    `moved` is what this implementation requests: f2_j is recomputed from the gathered row
    (node_attn.hip), so the 4K-byte f2 gather does not exist -- per edge 4 + D*s.
    `compulsory`: every table row read exactly once (SURVEY.md 8d honesty guard)."""
    if stats_row is None:
        from han_amd import ops
        stats_row = ops.STATS_ROW_BYTES
    if tag == "eval":
        per_row = D * 4 + K * 4 + 8                      # out row, f1_i, rowptr
        return {"algorithmic": nnz * (4 + D * s + K * 4) + n_rows * per_row,
                "moved": nnz * (4 + D * s) + n_rows * per_row,
                "compulsory": nnz * 4 + n_rows * (D * s + per_row)}
    if tag == "train":
        per_row = D * 4 + K * 4 + 8 + 2 * D * 4 + 2 * K * 4   # + pre, aggp rows and lse, tsum
        return {"algorithmic": nnz * (4 + D * s + K * 4) + n_rows * per_row,
                "moved": nnz * (4 + D * s) + n_rows * per_row,
                "compulsory": nnz * 4 + n_rows * (D * s + per_row)}
    if tag == "bwd_cols":
        # transposed-graph gather: per edge rowidx 4 + g row D*s + the stats record; per source
        # row H_j (D*s) + f2, df1 (4K each) + colptr 8 in, dH (4D) + df2 (4K) out
        per_row = D * s + D * 4 + 3 * K * 4 + 8
        b = nnz * (4 + D * s + stats_row) + n_rows * per_row
        return {"algorithmic": b, "moved": b,
                "compulsory": nnz * 4 + n_rows * (D * s + stats_row + per_row)}
    raise ValueError(tag)


K2_KERNEL_NAME = {"eval": "node_attn_fwd_kernel<FP=8,TRAIN=0> (K2 forward, eval)",
                  "train": "node_attn_fwd_kernel<FP=8,TRAIN=1,FAST> (K2 forward, training step)",
                  "bwd_cols": "node_attn_bwd_cols_kernel<FP=8,FAST> (K2 backward, transposed-graph gather)"}


def k2_rooflines(timing, esz, regime, cache_served=False):
    """Per K2 kernel: mean launch duration (HIP events on the launch stream) -> GB/s of moved
    bytes against the 8 TB/s HBM peak.  Returns {tag: dict} and the tag with the largest total time."""
    ms, shape = {}, {}
    for tag, e0, e1, nr, nnz in timing:
        ms.setdefault(tag, []).append(e0.elapsed_time(e1))
        shape[tag] = (nr, nnz)
    out, total = {}, {}
    for tag, v in ms.items():
        nr, nnz = shape[tag]
        avg = sum(v) / len(v)
        total[tag] = sum(v)
        b = k2_bytes(tag, nr, nnz, s=esz)
        ach = b["moved"] / (avg * 1e-3) / 1e9
        out[tag] = {"bound": "hbm", "kernel": K2_KERNEL_NAME[tag], "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "frac_of_measured_copy_rate": round(ach / HBM_COPY_GBS, 4),
                    "bytes_model": "moved = bytes the kernel requests (f2_j recomputed from the gathered row, "
                                   "no f2 gather); SURVEY.md 8d's 292 B/edge figure is algorithmic_survey_8d",
                    "moved_bytes_per_launch": b["moved"],
                    "algorithmic_survey_8d_bytes_per_launch": b["algorithmic"],
                    "algorithmic_survey_8d_GBs": round(b["algorithmic"] / (avg * 1e-3) / 1e9, 1),
                    "compulsory_bytes_per_launch": b["compulsory"],
                    "avg_launch_ms": round(avg, 4), "launches_timed": len(v),
                    "total_ms_in_timed_region": round(total[tag], 3), "rows": nr, "edges": nnz,
                    "regime": regime}
        if cache_served:
            # gather table <= ~2x the 256 MiB Infinity Cache: bytes / time is a fabric / cache-path rate, priced
            # against the HBM peak only for scale -- the HBM statement of the SAME kernel is `hbm_frac`
            # (filled in from roofline_hbm_regime), never `frac`
            out[tag]["bound"] = "fabric / infinity cache"
            out[tag]["hbm_frac"] = None
        if ach > HBM_PEAK_GBS:
            # e.g. power-law graphs: a few hub rows take most of the gathers and stay in the L2s
            out[tag]["bound"] = "l2 / infinity cache"
            out[tag]["note"] = ("requested bytes per second exceed the HBM peak: the gather is served from the "
                                "caches (hot rows), this is not an HBM rate")
    dom = max(total, key=total.get) if total else None
    return out, dom


def _src_sha():
    """sha256 (first 16 hex) of the K2 sources: ties a committed PMC traffic figure to the kernels it measured."""
    import hashlib
    h = hashlib.sha256()
    for f in ("han_amd/csrc/node_attn.hip", "han_amd/csrc/han_common.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def static_traffic(workload, world, table_dtype):
    """roofline.traffic: HBM-side bytes per launch from the PMC counters.  PMC collection needs rocprofv3
    around the process (tools/pmc_traffic.sh), so a plain bench run can only quote the committed
    figure -- and only when it was taken on these exact kernel sources; otherwise null."""
    tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None, None
    if tj.get("workload") != workload or tj.get("n_gpus", 1) != world or table_dtype != "f32":
        return None, None
    if tj.get("kernel_src_sha") != _src_sha():
        return None, ("profiles/k2_traffic.json was measured on different kernel sources "
                      f"({tj.get('kernel_src_sha')}); re-run tools/pmc_traffic.sh")
    per = {k: v.get("hbm_bytes_per_launch") for k, v in tj.get("detail", {}).items()}
    src = (f"profiles/k2_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
           f"(2*FETCH+WRITE)*1024, {tj.get('date', 'round ' + str(tj.get('round')))}, kernel sources {tj.get('kernel_src_sha')}); "
           "fabric-side counter: includes Infinity-Cache hits")
    return per, src


K2_PMC_NAMES = {"k2_fwd_eval": "node_attn_fwd_kernel<8, false, 1,", "k2_fwd_train": "node_attn_fwd_kernel<8, true, 1,",
                "k2_bwd_cols": "node_attn_bwd_cols_kernel<8, 1,"}


def live_traffic(args, timeout_s=150):
    """roofline.traffic measured in THIS run: two child runs of this script (one epoch of the same
    workload, no warm-up) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes with
    the kernel trace only, as MI355X_MICROARCH.md prescribes -- and bytes per launch = (2 * FETCH_SIZE +
    WRITE_SIZE) * 1024 (both counters are in KB; gfx950 tallies 128-B fetch requests as 64 B).  Returns
    ({k2 tag: bytes per launch}, source) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="han_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HAN_BENCH_DIAG")}
    env["TMPDIR"] = "/tmp"
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", c, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline", "--hbm-regime-nodes", "0", "--traffic", "off", "--skew-steps", "0",
                   "--workload", args.workload, "--table-dtype", args.table_dtype]
            if args.nodes:
                cmd += ["--nodes", str(args.nodes)]
            # own session: on a timeout the whole group goes (rocprofv3 AND the profiled bench.py child), so
            # that nothing of it is still on the GPU when the HBM-regime probe is timed afterwards
            pr = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            try:
                _, err = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                pr.wait()
                raise
            if pr.returncode != 0:
                return None, f"rocprofv3 --pmc {c} pass failed (rc {pr.returncode}): {err[-300:]}"
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                for tag, sub in K2_PMC_NAMES.items():
                    if sub in row["Kernel_Name"]:
                        vals[tag][row["Counter_Name"]].append(float(row["Counter_Value"]))
    except subprocess.TimeoutExpired:
        return None, f"rocprofv3 pass exceeded {timeout_s} s"
    except Exception as e:      # a profiler problem must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    per = {}
    for tag, d in vals.items():
        if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
            fetch = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
            write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            per[tag] = int((2 * fetch + write) * 1024)
    if not per:
        return None, "no K2 launches in the counter output"
    n_l = {t: len(vals[t]["FETCH_SIZE"]) for t in per}
    return per, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, kernel "
                 f"trace only) around one-epoch child runs of the same workload, mean over {n_l} launches, "
                 "(2*FETCH_SIZE + WRITE_SIZE)*1024 B with the gfx950 x2 on FETCH_SIZE (MI355X_MICROARCH.md); "
                 "fabric-side counter: includes Infinity-Cache hits")


def hbm_regime_probe(dev, n, table_dtype, steps, warmup=2, deg=50):
    """A second timed K2 launch set in the SAME run on a table far larger than the 256 MiB Infinity Cache
    (N = 10M rows: 2.56 GB fp32 / 1.28 GB bf16), so that bytes / time is a genuine HBM rate."""
    from han_amd import ops, synth
    g = synth.random_regular_graph(n, deg, 777, dev)
    gt = g.transpose()
    gen = torch.Generator(device=dev).manual_seed(11)
    tdt = torch.bfloat16 if table_dtype == "bf16" else torch.float32
    H = torch.randn((n, 64), device=dev, generator=gen).to(tdt)
    f1 = torch.randn((n, 8), device=dev, generator=gen)
    a1 = torch.randn((8, 8), device=dev, generator=gen) * 0.3
    a2 = torch.randn((8, 8), device=dev, generator=gen) * 0.3
    b2 = torch.zeros(8, device=dev)
    c = torch.zeros(64, device=dev)
    f2 = (H.float().view(n, 8, 8) * a2[None]).sum(-1)
    dOut = torch.randn((n, 64), device=dev, generator=gen)
    out = torch.empty((n, 64), device=dev)
    ops.K2_TIMING = None
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize()
            ops.K2_TIMING = []
        ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out)
        _, saved = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6,
                                     seed=1000 + it)
        pre, lse, aggp, tsum = saved
        gs, df1, _ = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, f1, lse, c, table_dtype=tdt)
        ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=1000 + it)
        del saved, pre, lse, aggp, tsum, gs, df1
    torch.cuda.synchronize()
    timing, ops.K2_TIMING = ops.K2_TIMING, None
    roofs, _ = k2_rooflines(timing, 2 if table_dtype == "bf16" else 4,
                            f"HBM-served: one meta-path of N={n} rows, deg {deg}, "
                            f"{n * 64 * (2 if table_dtype == 'bf16' else 4) / 1e9:.2f} GB gather table "
                            f"(10x the 256 MiB Infinity Cache)")
    return roofs


def skew_variant(dev, args, tdt, esz, warmup=2):
    """SURVEY.md 8d's second line: the same model on the power-law variant of the workload (alpha = 2.1, the same
    N, E, F, P; rows longer than 1024 edges are cut into chunks) -- graph generation + `warmup` + --skew-steps
    epochs in this process, after the headline's timed region.  Hub rows stay in the L2s, so a K2 rate above the
    HBM peak is labelled a cache rate (k2_rooflines)."""
    from han_amd import ops, rng, synth
    from han_amd.gat import HeteGAT_multi
    from han_amd.trainer import HANTrainer
    t_gen = time.perf_counter()
    wl = synth.make_workload("syn-1m-skew", device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    if tdt == torch.bfloat16:
        wl["x"] = wl["x"].to(torch.bfloat16)
    rng.manual_seed(2024)
    model = HeteGAT_multi().build(wl["p"], wl["f"], wl["c"], (8,), (8, 1), 128, device=dev,
                                  generator=torch.Generator().manual_seed(0), table_dtype=tdt)
    max_deg = max(int(g.degrees().max()) for g in wl["graphs"])
    e_total = sum(g.nnz for g in wl["graphs"])
    tr = HANTrainer(model, [wl["x"]] * wl["p"], wl["graphs"], wl["labels"], wl["train_mask"], wl["val_mask"],
                    lr=0.005, l2_coef=0.001, attn_drop=0.6, ffd_drop=0.6)
    wl["graphs"] = None
    for _ in range(warmup):
        tr.epoch()
    torch.cuda.synchronize()
    ops.K2_TIMING = []
    t0 = time.perf_counter()
    for _ in range(args.skew_steps):
        last = tr.epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing, ops.K2_TIMING = ops.K2_TIMING, None
    roofs, dom = k2_rooflines(timing, esz, "power-law degrees: the hub rows are served from the L2s / Infinity Cache",
                              cache_served=True)
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "moved_bytes_per_launch", "note")
    tl, ta, vl, va = tr.reduce_metrics(*last)
    del tr, model
    torch.cuda.empty_cache()
    return {"workload": f"syn-1m-skew: N={wl['n']}, P={wl['p']}, E={e_total} (power-law degrees, alpha 2.1, max degree "
                        f"{max_deg}), F={wl['f']}", "value": round(args.skew_steps / dt, 4), "unit": "epochs/s",
            "ms_per_step": round(dt / args.skew_steps * 1e3, 3), "steps": args.skew_steps, "warmup": warmup,
            "graph_generation_s": round(t_gen, 2),
            "k2": {tag: {k: r[k] for k in keep if k in r} for tag, r in roofs.items()},
            "final": {"train_loss": round(tl, 5), "val_loss": round(vl, 5)}}


# --------------------------------------------------------------------------- CPU baseline
def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // per)
        except Exception:
            pass
    return n if quota is None else min(n, quota), n, quota


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


_MALLOC_TUNED = False


def _tune_host_malloc():
    """glibc returns every large tensor to the kernel on free (mmap threshold), so a torch-CPU training step
    that allocates ~100 fresh 50 MB temporaries per epoch spends most of its time in page faults -- serial
    kernel work that no thread count helps (measured here: 31 s per epoch at N = 50 000 with 8 threads, 10.7 s
    with the two mallopt calls below, and the 1-thread / 8-thread ratio goes from 1.0 to 6.3).  TensorFlow, the
    reference's runtime, keeps its memory in a BFC arena; M_MMAP_MAX = 0 + no trimming is the closest glibc
    has.  Only the CPU-baseline leg (the last thing bench.py does) runs under it."""
    global _MALLOC_TUNED
    if _MALLOC_TUNED:
        return True
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        ok = libc.mallopt(-4, 0) == 1 and libc.mallopt(-1, 0x7FFFFFFF) == 1      # M_MMAP_MAX, M_TRIM_THRESHOLD
    except Exception:
        ok = False
    _MALLOC_TUNED = ok
    return ok


class _CpuEpochs:
    """The torch-CPU restatement of one reference epoch on a sample of the workload (same degree / feature
    width / meta-path count, fewer nodes); the dropout draws are part of the timed work, as in TF."""

    def __init__(self, workload_name, sample_n, seed=1234, dense=False):
        from han_amd import synth
        from oracle import han_oracle as ho
        from oracle import han_oracle_torch as ht
        import numpy as np
        self.ht, self.dense, self.n = ht, dense, sample_n
        wl = synth.make_workload(workload_name, device="cpu", seed=seed, n_override=sample_n)
        params = ho.init_params(np.random.default_rng(0), wl["p"], wl["f"], wl["c"])
        self.bp = ht.to_batched(params, dtype=torch.float32)
        self.state = ht.new_adam_state(self.bp)
        if dense:       # the reference's own form: materialised N x N additive masks (utils/layers.py:26-27)
            self.graphs = [g.to_bias()[0] for g in wl["graphs"]]
        else:
            self.graphs = [(g.rowptr, g.colidx) for g in wl["graphs"]]
        self.nnz = [g.nnz for g in wl["graphs"]]
        self.onehot = torch.nn.functional.one_hot(wl["labels"].long(), wl["c"]).float()
        self.xs = [wl["x"]] * wl["p"]
        self.f = wl["f"]
        self.train_mask, self.val_mask = wl["train_mask"].bool(), wl["val_mask"].bool()
        self.gens = [torch.Generator().manual_seed(p) for p in range(wl["p"])]

    def _masks_of(self, p):
        n, f, gen = self.n, self.f, self.gens[p]
        coef_shape = (8, n, n) if self.dense else (self.nnz[p], 8)
        return {"seq": (torch.rand((8, n, f), generator=gen) < 0.4).float(),
                "coef": (torch.rand(coef_shape, generator=gen) < 0.4).float(),
                "fts": (torch.rand((n, 64), generator=gen) < 0.4).float()}

    def one_epoch(self, threads):
        if threads > 1:      # torch's CPU generator is serial: one stream per meta-path, drawn concurrently
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(self.gens)) as ex:
                mk = list(ex.map(self._masks_of, range(len(self.gens))))
        else:
            mk = [self._masks_of(p) for p in range(len(self.gens))]
        self.ht.train_epoch(self.xs, self.graphs, self.bp, self.state, self.onehot, self.train_mask, self.val_mask,
                            keep=0.4, masks=mk, dense=self.dense)

    def time(self, threads, min_epochs, budget_s, warmup=True):
        """Median seconds per epoch with `threads` threads."""
        torch.set_num_threads(threads)
        if warmup:
            self.one_epoch(threads)
        times = []
        t_start = time.perf_counter()
        while len(times) < min_epochs or (time.perf_counter() - t_start < budget_s and len(times) < 3 * min_epochs):
            t0 = time.perf_counter()
            self.one_epoch(threads)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > 3 * budget_s:
                break
        times.sort()
        return times[len(times) // 2], len(times)


def cpu_baseline(workload_name, n_full, sample_n):
    """The reference algorithm (CSR restatement, torch CPU fp32) on a bounded sample of the same
    workload; epochs/s is scaled by sample_n / n_full (every term of the algorithm is linear in N at fixed
    degree).  Timed with all usable host cores and with ONE thread ON THE SAME SAMPLE;
    `value` is the all-cores figure, `cores` the threads it used, `cores_effective` the measured speed-up
    over one thread (what the cores really buy)."""
    cores, affinity, quota = host_cores()
    prev = torch.get_num_threads()
    tuned = _tune_host_malloc()
    run = _CpuEpochs(workload_name, sample_n)
    t_all, reps_all = run.time(cores, 3, 12.0)
    t_one, reps_one = run.time(1, 1, 1.0, warmup=False)       # the all-cores epochs were the warm-up
    out = {"value": (1.0 / t_all) * sample_n / n_full, "unit": "epochs/s", "cores": cores,
           "cores_effective": round(t_one / t_all, 2), "kind": "port",
           "sample": f"{workload_name} shape at N={sample_n} (deg/F/P unchanged), median of {reps_all} epochs "
                     f"({t_all:.2f} s each) after one warm-up, {cores} threads, scaled by {sample_n}/{n_full} "
                     f"(extrapolation factor {n_full / sample_n:.0f}x); the same sample with ONE thread: {t_one:.2f} s "
                     f"per epoch; torch-CPU fp32 CSR restatement of the reference (not TensorFlow)",
           "all_cores": {"threads": cores, "sample_n": sample_n, "epoch_s": round(t_all, 3), "epochs_timed": reps_all,
                         "epochs_per_s_scaled": (1.0 / t_all) * sample_n / n_full},
           "one_thread": {"threads": 1, "sample_n": sample_n, "epoch_s": round(t_one, 3), "epochs_timed": reps_one,
                          "epochs_per_s_scaled": (1.0 / t_one) * sample_n / n_full},
           "host_allocator": ("glibc malloc with M_MMAP_MAX = 0 and trimming off (a caching arena, like TF's BFC "
                              "allocator): without it the port is page-fault-bound and no thread count helps"
                              if tuned else "glibc default (mallopt unavailable)"),
           "host": {"cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "affinity": affinity,
                    "cgroup_quota_cores": quota}}
    if workload_name == "acm-like":
        # the reference's actual algorithm on its own dataset shape: dense N x N masks, full size
        t_d, reps_d = _CpuEpochs(workload_name, n_full, dense=True).time(cores, 2, 20.0)
        out["dense_reference_form"] = {"threads": cores, "sample_n": n_full, "epoch_s": round(t_d, 3),
                                       "epochs_timed": reps_d, "epochs_per_s": 1.0 / t_d,
                                       "note": "materialised N x N logits + additive -1e9 mask, op for op with "
                                               "utils/layers.py:20-35 (full size, no extrapolation)"}
    torch.set_num_threads(prev)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--replicate", default="auto", choices=["auto", "all", "eval", "none"],
                    help="multi-GPU: which forward passes project the whole H table on every rank instead of "
                         "exchanging it (auto: both up to 4 ranks, the eval forward beyond)")
    ap.add_argument("--traffic", choices=("live", "static", "off"), default="live",
                    help="roofline.traffic: live = PMC passes (rocprofv3 child runs of one epoch) in this run, "
                         "N = 1 on the syn-1m family; static = the committed profiles/k2_traffic.json when it "
                         "matches the kernel sources; off = null")
    ap.add_argument("--workload", default="syn-1m")
    ap.add_argument("--nodes", type=int, default=0, help="override N (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the epoch into a hipGraph and replay it (single GPU; small graphs)")
    ap.add_argument("--cpu-sample", type=int, default=50000,
                    help="nodes of the CPU-baseline sample (timed with all usable cores AND with one thread)")
    ap.add_argument("--hbm-regime-nodes", type=int, default=10_000_000,
                    help="rows of the extra single-meta-path table on which the K2 kernels are re-timed in the "
                         "HBM-served regime after the timed region (0 = skip; single GPU only)")
    ap.add_argument("--skew-steps", type=int, default=5,
                    help="epochs timed on the power-law variant of the headline workload (syn-1m-skew, same N / E / F / P) "
                         "in the same process after the timed region; reported as `skew` (0 = skip; N = 1, syn-1m only)")
    ap.add_argument("--masked-backward", action="store_true",
                    help="ALSO time the opt-in masked backward (HANTrainer.set_masked_backward: destinations outside the "
                         "train mask are skipped in the transposed-graph pass and, under a partition, only the live rows "
                         "of the backward table travel) after the headline region; reported as `masked_backward`, never "
                         "as `value`")
    ap.add_argument("--reorder", choices=("none", "bfs"), default="none",
                    help="locality pass (han_amd.reorder): breadth-first relabelling of the nodes before training "
                         "(single GPU; the pass itself is timed separately and reported)")
    ap.add_argument("--overlap-eval", nargs="?", const="branch", default=None, choices=("branch", "sections"),
                    help="with --graph: the eval forward of epoch k - 1 as a second branch of the captured epoch beside "
                         "the training step of epoch k (HANTrainer(overlap_eval=True); same numbers, the validation "
                         "pair one call late)")
    ap.add_argument("--side-stream", action="store_true",
                    help="HANTrainer(side_stream=True): the backward's dW of meta-path p on a second stream beside the "
                         "gather of meta-path p + 1 (+1.3-2.5 %% epochs/s at SYN-1M; off by default because the gather's "
                         "launch time -- the line's roofline -- then includes that company; the line carries "
                         "roofline.alone beside it)")
    ap.add_argument("--no-dense", action="store_true",
                    help="keep small dense graphs on the lean CSR kernels (measurements: the matrix-pipe K2 form off)")
    ap.add_argument("--table-dtype", choices=("f32", "bf16"), default="f32",
                    help="storage of X and of the H/g gather tables (bf16 = the 10M-node config's "
                         "'bf16 feats'); accumulation is fp32 either way")
    args = ap.parse_args()
    if os.environ.get("HAN_DEBUG_HANG"):      # dump every thread's Python stack periodically (rehearsal debugging)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["HAN_DEBUG_HANG"]), repeat=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # N > 1 without RANK was turned into N ranks by launch_ranks_if_needed above
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # a tree without the built library (a fresh checkout: the .so is git-ignored): compile it once,
    # before anything touches the GPU -- local rank 0 builds, the others wait for the file
    from han_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        if local_rank == 0:
            _lib.build()
        t_wait = time.time()
        while not os.path.exists(_lib.LIB_PATH):
            if time.time() - t_wait > 600:
                raise SystemExit(f"{_lib.LIB_PATH} was not built")
            time.sleep(1.0)
    # HAN_SHARE_GPU=1 + HAN_DIST_BACKEND=gloo: rehearse the N-rank partition with all ranks on
    # GPU 0 of a one-GPU box (collectives staged through the host; timings are meaningless)
    if os.environ.get("HAN_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HAN_FORCE_COLLECTIVES=1 rehearses the RCCL path on a 1-rank group (single-GPU box)
    use_dist = world > 1 or (os.environ.get("HAN_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("HAN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from han_amd import ops, rng, synth
    from han_amd.dist import NodePartition
    from han_amd.gat import HeteGAT_multi
    from han_amd.trainer import HANTrainer

    if args.no_dense:
        ops.DENSE = False
    rng.manual_seed(2024)
    cfg_n = args.nodes or synth.CONFIGS[args.workload]["n"]
    part = NodePartition(cfg_n, rank, world) if use_dist else None
    # under a partition every rank generates ONLY its own rows (features, labels, masks, graph rows
    # with global column ids): setup memory is global/world + the exchange tables, by construction
    wl = synth.make_workload(args.workload, device=dev, n_override=args.nodes or None,
                             rows=(part.row_start, part.row_end) if part is not None else None)
    n, p = wl["n"], wl["p"]
    reorder_info = None
    if args.reorder != "none":
        if part is not None:
            raise SystemExit("--reorder is a single-process pass over the whole graph")
        from han_amd import reorder
        torch.cuda.synchronize()
        t_r = time.perf_counter()
        halo_before = reorder.halo_fraction(wl["graphs"], 8)
        rel = reorder.relabel(wl, args.reorder)
        wl = rel.wl
        torch.cuda.synchronize()
        reorder_info = {"method": args.reorder, "seconds": round(time.perf_counter() - t_r, 2),
                        "halo_fraction_8_ranks_before": [round(h, 4) for h in halo_before],
                        "halo_fraction_8_ranks_after": [round(h, 4) for h in reorder.halo_fraction(wl["graphs"], 8)]}
    gen = torch.Generator().manual_seed(0)
    tdt = torch.bfloat16 if args.table_dtype == "bf16" else torch.float32
    model = HeteGAT_multi().build(p, wl["f"], wl["c"], (8,), (8, 1), 128, device=dev, generator=gen,
                                  table_dtype=tdt)
    if tdt == torch.bfloat16:
        wl["x"] = wl["x"].to(torch.bfloat16)
    e_per_path = [int(g.nnz) for g in wl["graphs"]]          # (this rank's rows; quoted only when they differ)
    e_local = torch.tensor([float(sum(g.nnz for g in wl["graphs"]))], dtype=torch.float64)
    if part is not None:
        e_local = e_local.to(dev) if dist.get_backend() == "nccl" else e_local
        dist.all_reduce(e_local)
    e_global = int(e_local.item())
    # Replicated projection (han_amd/dist.py:replication_policy): the forward passes it names project the
    # whole H table on every rank from the features of ALL rows (1 KB per row, generated by every rank
    # from the same block seeds) instead of receiving (G-1)/G of it over the xGMI links; the graph, the
    # labels and every other per-row tensor stay rank-local.
    x_full = None
    if part is not None:
        from han_amd.dist import replication_policy
        if replication_policy(world, args.replicate):
            x_full = synth.features(args.workload, device=dev, n_override=args.nodes or None)
            if tdt == torch.bfloat16:
                x_full = x_full.to(torch.bfloat16)
    trainer = HANTrainer(model, [wl["x"]] * p, wl["graphs"], wl["labels"], wl["train_mask"],
                         wl["val_mask"], lr=0.005, l2_coef=0.001, attn_drop=0.6, ffd_drop=0.6,
                         part=part, use_graph=args.graph and part is None, graphs_local=part is not None,
                         xs_full=[x_full] * p if x_full is not None else None, replicate=args.replicate,
                         side_stream=bool(args.side_stream),
                         overlap_eval=(args.overlap_eval if (args.overlap_eval and args.graph and part is None) else False))
    exchange = None
    if part is not None:
        plans = model.halo_plans[0]
        rep = sorted(trainer.replicate)
        ag = {(): "all-gather",
              ("eval",): "all-gather (training forward + backward); H projected on every rank in the eval forward",
              ("eval", "train"): "all-gather (backward only); H projected on every rank in both forwards"}[tuple(rep)]
        exchange = ["halo %.1f%% of the remote rows" % (100.0 * pl.halo_fraction) if pl is not None else ag
                    for pl in plans]
    # which meta-paths run K2 in the dense (bit mask + fp32 MFMA) form: small graphs at least half full
    dense_paths = None
    if part is None and n <= 16384:
        probe = torch.empty((n, 64), device=dev)
        dense_paths = [bool(ops._use_dense(g, probe, 8, 8)) for g in trainer.graphs]
        del probe
    trainer_replicate_info = trainer.replicate_info
    wl["graphs"] = None
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.epoch()
    barrier()
    comm = None
    if part is not None:
        from han_amd.dist import CommStats
        comm = part.comm = CommStats()
    ops.K2_TIMING = None if trainer.use_graph else []      # a replayed graph records no events
    ms0 = torch.cuda.memory_stats(dev) if os.environ.get("HAN_BENCH_DIAG") else None
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = trainer.epoch()
    barrier()
    dt = time.perf_counter() - t0
    if ms0 is not None:      # allocator activity inside the timed region (a hipMalloc / hipFree there is a stall)
        ms1 = torch.cuda.memory_stats(dev)
        keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")
        print("[diag] allocator in the timed region:", {k: ms1.get(k, 0) - ms0.get(k, 0) for k in keys},
              "reserved GB %.1f allocated peak GB %.1f" % (ms1["reserved_bytes.all.current"] / 1e9,
                                                            ms1["allocated_bytes.all.peak"] / 1e9), file=sys.stderr)
    timing, ops.K2_TIMING = ops.K2_TIMING or [], None
    # the training step runs dW of meta-path p on a second stream beside the backward gather of meta-path p + 1
    # (HANTrainer(side_stream=...)): the gather's launch time in the timed region includes that company.  Two more
    # epochs on one stream, outside the timed region, give the same kernels' launch times ALONE.
    timing_alone = None
    if model.side_stream is not None and not trainer.use_graph:
        side, model.side_stream = model.side_stream, None
        ops.K2_TIMING = []
        for _ in range(2):
            trainer.epoch()
        torch.cuda.synchronize(dev)
        timing_alone, ops.K2_TIMING = ops.K2_TIMING, None
        model.side_stream = side
    comm_info = None
    if part is not None:
        part.comm = None
        # max over ranks of the time the compute stream waited for collectives, per step
        cw = torch.tensor([comm.wait_ms() / args.steps], dtype=torch.float64,
                          device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(cw, op=dist.ReduceOp.MAX)
        comm_info = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                     "exchange_bytes_received": int(comm.bytes_received // args.steps),
                     "exchanges_per_step": comm.exchanges // args.steps,
                     "grad_allreduce_bytes": int(comm.allreduce_bytes // args.steps),
                     "comm_wait_ms": round(float(cw.item()), 3),
                     "comm_note": "per rank and step (one epoch): exchange_bytes_received = table rows this rank "
                                  "received (all-gather blocks of the other ranks / halo rows); comm_wait_ms = HIP "
                                  "events on the compute stream around every wait for a collective (+ host time of "
                                  "host-staged gloo collectives), max over ranks"}
    use_graph = trainer.use_graph
    overlap_form = trainer.overlap_form if trainer.overlap_eval else None      # (the trainer is released before the line is built)
    masked_info = None
    if args.masked_backward and not use_graph:
        # the extra, never the headline: same model state, same steps, the backward restricted to the live rows
        trainer.set_masked_backward(True)
        for _ in range(max(args.warmup, 1)):
            trainer.epoch()
        barrier()
        t0m = time.perf_counter()
        for _ in range(args.steps):
            trainer.epoch()
        barrier()
        dtm = time.perf_counter() - t0m
        if use_dist:
            tm = torch.tensor([dtm], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dtm = float(tm.item())
        rb = ops.gs_row_bytes(8, 8, tdt)
        plans_m = trainer._masked_plans
        masked_info = {"value": round(args.steps / dtm, 4), "unit": "epochs/s",
                       "ms_per_step": round(dtm / args.steps * 1e3, 3),
                       "live_fraction": round(float(trainer.train_mask.float().mean()), 4),
                       "bytes_on_wire": {"backward_table_per_rank_per_step_masked": int(sum(pl.rows_on_wire for pl in plans_m) * rb),
                                         "backward_table_per_rank_per_step_full": int((world - 1) * (part.shard if part is not None else 0) * rb * p)},
                       "note": "opt-in: bit-identical results (dead entries of the transposed graph skipped in place); "
                               "the headline `value` is the full pass"}
        trainer.set_masked_backward(False)
    if use_dist:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tl, ta, vl, va = trainer.reduce_metrics(*last)

    # --- K2 rooflines from the HIP events recorded inside the timed region ------------------
    esz = 2 if args.table_dtype == "bf16" else 4
    table_mb = n * 64 * esz / 1e6
    regime = (f"gather table {table_mb:.0f} MB per meta-path vs the 256 MiB Infinity Cache: "
              + ("largely cache-served -- an effective (fabric/cache-path) rate, NOT an HBM fraction; "
                 "see roofline_hbm_regime" if table_mb < 600 else "HBM-served"))
    cache_served = table_mb < 600
    roofs, dom = k2_rooflines(timing, esz, regime, cache_served=cache_served)
    traffic_per, traffic_src = None, None
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
    if args.traffic == "live" and profiled:
        args.traffic = "static"      # already running under a profiler: no nested rocprofv3
    if args.traffic == "live" and rank == 0 and world == 1 and not use_graph and args.workload.startswith("syn-1m"):
        traffic_per, traffic_src = live_traffic(args)
        if traffic_per is None:
            live_err = traffic_src
            traffic_per, traffic_src = static_traffic(args.workload, world, args.table_dtype)
            traffic_src = f"{traffic_src} [live PMC collection failed: {live_err}]" if traffic_src else \
                f"live PMC collection failed: {live_err}"
    elif args.traffic != "off":
        traffic_per, traffic_src = static_traffic(args.workload, world, args.table_dtype)
    tkey = {"eval": "k2_fwd_eval", "train": "k2_fwd_train", "bwd_cols": "k2_bwd_cols"}
    for tag, r in roofs.items():
        r["traffic"] = (traffic_per or {}).get(tkey[tag]) if traffic_per else None
        r["traffic_source"] = traffic_src
    hbm = None
    skew = None
    if rank == 0 and world == 1 and args.hbm_regime_nodes > 0 and not use_graph and table_mb < 600 \
            and args.workload.startswith("syn-1m"):      # the headline family; other workloads are not priced against HBM
        del trainer, model
        torch.cuda.empty_cache()
        hbm = hbm_regime_probe(dev, args.hbm_regime_nodes, args.table_dtype, min(max(args.steps, 5), 20))
        for tag, r in roofs.items():                     # the HBM fraction of the same kernel, same run
            if "hbm_frac" in r and tag in hbm:
                r["hbm_frac"] = hbm[tag]["frac"]
    if rank == 0 and world == 1 and args.skew_steps > 0 and not use_graph and args.workload == "syn-1m" \
            and not args.nodes:
        skew = skew_variant(dev, args, tdt, esz)

    if rank == 0:
        out = {
            # BASELINE.json's metric, verbatim; `value` is its epochs/s part, the node-attn GB/s part is `roofline`
            "metric": "full-graph train epochs/sec + node-attn HBM GB/s, 8-head HAN at 1/2/4/8 GPUs",
            "metric_note": "value = train epochs/s (1 epoch = fwd+bwd+Adam step with dropout 0.6/0.6 + one eval "
                           "forward, ex_acm3025.py:171-218); node-attn GB/s = roofline.achieved",
            "value": round(args.steps / dt, 4), "unit": "epochs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            **(comm_info or {}),
            "dtype": "f32" if args.table_dtype == "f32" else "bf16 storage / f32 accumulate",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: N={n} nodes, P={p} meta-paths, "
                                   f"E={e_global} " + (f"{e_per_path} " if len(set(e_per_path)) > 1 and world == 1 else "") +
                                   f"edges, F={wl['f']}, K=8 heads x F'=8, A=128, C={wl['c']}",
                       "graph_recipe": ("SURVEY.md 8d: per row the self-loop + (deg-1) DISTINCT uniformly random "
                                        "off-diagonal neighbours, columns sorted per row; drawn with block-seeded torch "
                                        "generators on the device (so that N ranks generate exactly their own rows) "
                                        "instead of numpy default_rng(1234+p)")
                       if args.workload.startswith("syn-") else
                       "han_amd.synth: per meta-path a symmetric random graph + I with EXACTLY the data set's entry count "
                       "(SURVEY.md section 8; (nnz - N)/2 distinct pairs i < j drawn without replacement, mirrored; "
                       "ids ascending per row)",
                       "parallelism": f"node-partition x{world}" if world > 1 else
                       (("single GPU, epoch replayed from a hipGraph" +
                         ("; eval forward of epoch k-1 runs as a branch beside the training step of epoch k "
                          f"(overlap_eval = {overlap_form!r}: every epoch still holds one training step and one "
                          "eval forward)"
                          if overlap_form else "")) if use_graph else "single GPU"),
                       **({"captured_epoch": {"c_abi_calls": getattr(trainer, "graph_abi_calls", None),
                                              "kernel_nodes_per_epoch": "profiles/r04_*_like_graph_kernel_stats.csv "
                                                                        "(calls / 53 epochs)"}} if use_graph else {}),
                       **({"k2_dense_form": dense_paths} if dense_paths is not None else {}),
                       **({"exchange": exchange} if exchange is not None else {}),
                       **({"replication": trainer_replicate_info} if trainer_replicate_info else {}),
                       **({"reorder": reorder_info} if reorder_info is not None else {}),
                       "dropout": "0.6/0.6 (train step)", "optimizer": "TF-form Adam lr 0.005, L2 0.001"},
            "final": {"train_loss": round(tl, 5), "train_acc": round(ta, 5),
                      "val_loss": round(vl, 5), "val_acc": round(va, 5),
                      **({"val_of": "the parameters before the last step (overlap_eval)"} if overlap_form else {})},
        }
        if dom is not None:
            # the K2 kernel with the largest total time in the timed region
            out["roofline"] = dict(roofs[dom], dominant_by="total time among the K2 kernels in the timed region")
            out["roofline_k2_all"] = roofs
            if timing_alone:
                alone = k2_rooflines(timing_alone, esz, regime, cache_served=cache_served)[0]
                for key, r in roofs.items():
                    if key in alone:
                        r["alone"] = {k: alone[key][k] for k in ("avg_launch_ms", "achieved", "frac")}
                out["roofline"]["alone"] = roofs[dom]["alone"]
                out["roofline"]["co_scheduled"] = (
                    "the training step runs K1's dW of meta-path p on a second stream beside the backward gather of "
                    "meta-path p + 1 (3 of 4 launches): avg_launch_ms / achieved / frac are this kernel's launches in the "
                    "timed region, WITH that company; `alone` = the same launches in two further epochs of this run "
                    "on one stream (the default, without --side-stream, times the whole line that way)")
        else:
            out["roofline"] = None
            out["roofline_note"] = ("no per-kernel HIP events in this run (an epoch replayed from a hipGraph "
                                    "records none); run without --graph for the K2 rooflines")
        if masked_info is not None:
            out["masked_backward"] = masked_info
        if skew is not None:
            # "a second, skewed variant must be reported beside it" (real meta-path graphs are skewed)
            out["skew"] = skew
        if hbm is not None:
            # the >= 50 % HBM target of BASELINE.json is read off THIS object: same kernels, same run,
            # a table 10x larger than the Infinity Cache
            out["roofline_hbm_regime"] = hbm
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, n, min(args.cpu_sample, n))
            out["cpu_baseline"] = cb
            out["vs_cpu_baseline"] = round(out["value"] / cb["value"], 1)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
