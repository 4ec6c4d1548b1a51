#!/usr/bin/env python3
"""Headline benchmark: full-graph HAN train epochs/s + node-attention (K2) bandwidth.

    python bench.py --gpus N --steps K --warmup W [--workload syn-1m]

One "step" is one reference epoch (ex_acm3025.py:171-218): one full-graph
fwd+bwd+Adam step with dropout 0.6/0.6 plus one eval forward.  The workload is
BASELINE.json's configs[3] (SYN-1M: 1M nodes, 4 meta-paths, deg 50, 256-d feats,
8 heads x 8), synthetic data, random-init weights, node-partitioned over N GPUs
(strong scaling: the graph is fixed, each rank owns N/G rows).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel -- the K2
node-attention forward (eval instantiation) -- from ALGORITHMIC bytes
(E*(4+256+4K) + N*(256+4K+8) per launch) over its mean launch
duration measured with HIP events on the launch stream inside the timed region.
`cpu_baseline` is the torch-CPU port of the reference algorithm
(oracle/han_oracle_torch.py, CSR form) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def k2_algorithmic_bytes(n_rows, nnz, K=8, D=64, s=4, train=False):
    """the gathered row is s bytes per element (4 fp32 / 2 bf16); the
    output row is always written in fp32."""
    per_edge = 4 + D * s + K * 4
    per_row = D * 4 + K * 4 + 8
    if train:                       # + pre, aggp rows and lse, tsum
        per_row += 2 * D * 4 + 2 * K * 4
    return nnz * per_edge + n_rows * per_row


def cpu_baseline(workload_name, n_full, sample_n, seed=1234):
    """The reference algorithm (CSR restatement, torch CPU fp32, all cores) on a
    bounded sample: same degree / feature width / meta-path count, fewer nodes.
    epochs/s is scaled by sample_n / n_full (every term of the algorithm is
    linear in N at fixed degree)."""
    from han_amd import synth
    from oracle import han_oracle as ho
    from oracle import han_oracle_torch as ht
    import numpy as np
    wl = synth.make_workload(workload_name, device="cpu", seed=seed, n_override=sample_n)
    rng = np.random.default_rng(0)
    params = ho.init_params(rng, wl["p"], wl["f"], wl["c"])
    bp = ht.to_batched(params, dtype=torch.float32)
    state = ht.new_adam_state(bp)
    graphs = [(g.rowptr, g.colidx) for g in wl["graphs"]]
    onehot = torch.nn.functional.one_hot(wl["labels"].long(), wl["c"]).float()
    xs = [wl["x"]] * wl["p"]
    g = torch.Generator().manual_seed(0)

    def one_epoch():
        mk = []                  # the dropout draws are part of the timed work, as in TF
        for (rp, ci) in graphs:
            mk.append({"seq": (torch.rand((8, sample_n, wl["f"]), generator=g) < 0.4).float(),
                       "coef": (torch.rand((ci.numel(), 8), generator=g) < 0.4).float(),
                       "fts": (torch.rand((sample_n, 64), generator=g) < 0.4).float()})
        ht.train_epoch(xs, graphs, bp, state, onehot, wl["train_mask"].bool(), wl["val_mask"].bool(),
                       keep=0.4, masks=mk)

    one_epoch()                  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or time.perf_counter() - t0 < 10.0:
        one_epoch()
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = (time.perf_counter() - t0) / reps
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": (1.0 / dt) * sample_n / n_full, "unit": "epochs/s", "cores": torch.get_num_threads(),
            "kind": "port", "host": {"cpu_model": model, "os_cpu_count": os.cpu_count(),
                                     "torch_threads": torch.get_num_threads()},
            "sample": f"{workload_name} shape at N={sample_n} (deg/F/P unchanged), {reps} epochs of "
                      f"{dt:.2f} s, scaled by {sample_n}/{n_full}; torch-CPU fp32 CSR restatement "
                      f"of the reference (not TensorFlow)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="syn-1m")
    ap.add_argument("--nodes", type=int, default=0, help="override N (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the epoch into a hipGraph and replay it (single GPU; small graphs)")
    ap.add_argument("--cpu-sample", type=int, default=5000)
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
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    rng.manual_seed(2024)
    wl = synth.make_workload(args.workload, device=dev, n_override=args.nodes or None)
    n, p = wl["n"], wl["p"]
    part = NodePartition(n, rank, world) if use_dist else None
    gen = torch.Generator().manual_seed(0)
    tdt = torch.bfloat16 if args.table_dtype == "bf16" else torch.float32
    model = HeteGAT_multi().build(p, wl["f"], wl["c"], (8,), (8, 1), 128, device=dev, generator=gen,
                                  table_dtype=tdt)
    if tdt == torch.bfloat16:
        wl["x"] = wl["x"].to(torch.bfloat16)

    def loc(t):
        return part.local_rows(t).contiguous() if part is not None else t

    x_local = loc(wl["x"])
    e_global = sum(g.nnz for g in wl["graphs"])
    trainer = HANTrainer(model, [x_local] * p, wl["graphs"], loc(wl["labels"]), loc(wl["train_mask"]),
                         loc(wl["val_mask"]), lr=0.005, l2_coef=0.001, attn_drop=0.6, ffd_drop=0.6,
                         part=part, use_graph=args.graph and part is None)
    exchange = None
    if part is not None:
        plans = model.halo_plans[0]
        exchange = ["halo %.1f%% of the remote rows" % (100.0 * pl.halo_fraction) if pl is not None else "all-gather"
                    for pl in plans]
    if part is not None:
        wl["graphs"] = None          # the global graphs are no longer needed on this rank
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.epoch()
    barrier()
    ops.K2_TIMING = None if trainer.use_graph else []      # a replayed graph records no events
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = trainer.epoch()
    barrier()
    dt = time.perf_counter() - t0
    timing, ops.K2_TIMING = ops.K2_TIMING or [], None
    if use_dist:
        t = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tl, ta, vl, va = trainer.reduce_metrics(*last)

    # --- roofline of the dominant kernel (K2 forward, eval instantiation) -------------
    ms = {"eval": [], "train": [], "bwd_cols": []}
    shape = {}
    for tag, e0, e1, nr, nnz in timing:
        ms[tag].append(e0.elapsed_time(e1))
        shape[tag] = (nr, nnz)
    roof = None
    extra = {}
    if ms["eval"]:
        nr, nnz = shape["eval"]
        avg_ms = sum(ms["eval"]) / len(ms["eval"])
        esz = 2 if args.table_dtype == "bf16" else 4
        alg = k2_algorithmic_bytes(nr, nnz, s=esz)
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("n_gpus", 1) == world \
                        and args.table_dtype == "f32":
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": "node_attn_fwd_kernel<FP=8,TRAIN=0> (K2 forward)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg,
                # SURVEY.md 8d honesty guard: bytes if every table row were read exactly once
                # (E*4 + N*(2*D*s + 2*K*4 + 8)); the gap to `algorithmic` is the ~deg-fold re-read of H rows
                "compulsory_bytes_per_launch": nnz * 4 + nr * (64 * esz + 64 * 4 + 2 * 8 * 4 + 8),
                "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms["eval"])}
        if ms["train"]:
            nr, nnz = shape["train"]
            tavg = sum(ms["train"]) / len(ms["train"])
            talg = k2_algorithmic_bytes(nr, nnz, s=esz, train=True)
            extra["k2_train_fwd"] = {"avg_launch_ms": round(tavg, 4),
                                     "achieved_GBs": round(talg / (tavg * 1e-3) / 1e9, 1)}
        if ms["bwd_cols"]:
            # transposed-graph backward: per edge rowidx 4 + g row D*s + stats 128; per source
            # row H_j (D*s) + f2, df1 (K*4 each) + colptr 8 in, dH (D*4) + df2 (K*4) out
            nr, nnz = shape["bwd_cols"]
            bavg = sum(ms["bwd_cols"]) / len(ms["bwd_cols"])
            balg = nnz * (4 + 64 * esz + 128) + nr * (64 * esz + 64 * 4 + 3 * 8 * 4 + 8)
            extra["k2_bwd_cols"] = {"avg_launch_ms": round(bavg, 4),
                                    "achieved_GBs": round(balg / (bavg * 1e-3) / 1e9, 1)}

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
            "dtype": "f32" if args.table_dtype == "f32" else "bf16 storage / f32 accumulate",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: N={n} nodes, P={p} meta-paths, "
                                   f"E={e_global} "
                                   f"edges, F={wl['f']}, K=8 heads x F'=8, A=128, C={wl['c']}",
                       "parallelism": f"node-partition x{world}" if world > 1 else
                       ("single GPU, epoch replayed from a hipGraph" if trainer.use_graph else "single GPU"),
                       **({"exchange": exchange} if exchange is not None else {}),
                       "dropout": "0.6/0.6 (train step)", "optimizer": "TF-form Adam lr 0.005, L2 0.001"},
            "final": {"train_loss": round(tl, 5), "train_acc": round(ta, 5),
                      "val_loss": round(vl, 5), "val_acc": round(va, 5)},
        }
        if roof is not None:
            out["roofline"] = roof
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, n, min(args.cpu_sample, n))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
