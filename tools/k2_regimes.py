#!/usr/bin/env python3
"""K2 (node-attention forward, eval) alone at growing table sizes: from a table
that fits the 256 MiB Infinity Cache (N = 1M -> 256 MB) to tables far beyond it
(N = 10M -> 2.56 GB), deg 50, fp32.  Prints one JSON line per size with the
kernel time (HIP events on the launch stream, median of `reps`) and the
algorithmic bandwidth E*292 + N*296 bytes / time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import ops, synth  # noqa: E402


def train_variants(dev, sizes=(1_000_000, 4_000_000, 10_000_000)):
    if os.environ.get("N_ONLY"):
        sizes = (int(os.environ["N_ONLY"]),)
    """Training kernels (forward with both dropouts, transposed-graph backward) in the
    cache-resident and the HBM-served regime, fp32 and bf16 tables."""
    for n in sizes:
        g = synth.random_regular_graph(n, 50, 1234, dev)
        gt = g.transpose()
        gen = torch.Generator(device=dev).manual_seed(1)
        rnd = lambda *s: torch.randn(s, device=dev, generator=gen)
        X, W = rnd(n, 64), torch.eye(64, device=dev)
        a1, a2, b1, b2, c = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1, rnd(64) * 0.1
        out = torch.empty((n, 64), device=dev)
        dOut = rnd(n, 64)
        for tdt in (torch.float32, torch.bfloat16):
            H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3, table_dtype=tdt)
            tag = "f32" if tdt == torch.float32 else "bf16"

            def timeit(fn, reps=5):
                fn(); fn()
                ts = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return sorted(ts)[len(ts) // 2]
            te = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out))
            if "--deep" in sys.argv and tdt == torch.bfloat16:      # same process, same box: 4 against 8 steps in flight
                ops.K2_DEEP = True
                te8 = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out))
                ops.K2_DEEP = False
                te4 = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out))
                print(json.dumps({"N": n, "tables": tag, "bf16_eval_forward_ms": {"4_steps_in_flight": [round(te, 3), round(te4, 3)],
                                                                                   "8_steps_in_flight": round(te8, 3)}}), flush=True)
            tt = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6,
                                                  fts_drop=0.6, seed=3))
            _, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=3)
            gs, df1, dc = ops.node_attn_bwd_rows(dOut, sv[0], sv[2], sv[3], f1, sv[1], c, table_dtype=tdt)
            tb = timeit(lambda: ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6,
                                                       fts_drop=0.6, seed=3))
            # 128-B lines requested per edge: the gathered row (2 fp32 / 1 bf16) + in the backward the line with
            # the (f1, lse, s) records; DESIGN.md section 3 (K2): time ~ lines per edge, not bytes
            lf, lb = (2, 3) if tdt == torch.float32 else (1, 2)
            e = g.nnz
            print(json.dumps({"N": n, "tables": tag,  "fwd_eval_ms": round(te, 3), "fwd_train_ms": round(tt, 3),
                              "bwd_cols_ms": round(tb, 3), "lines_per_edge": {"fwd": lf, "bwd": lb},
                              "ns_per_1000_lines": {"fwd_eval": round(te * 1e6 / (e * lf) * 1e3, 2),
                                                    "fwd_train": round(tt * 1e6 / (e * lf) * 1e3, 2),
                                                    "bwd_cols": round(tb * 1e6 / (e * lb) * 1e3, 2)},
                              "line_TBs": {"fwd_eval": round(e * lf * 128 / te / 1e9, 2),
                                           "fwd_train": round(e * lf * 128 / tt / 1e9, 2),
                                           "bwd_cols": round(e * lb * 128 / tb / 1e9, 2)}}), flush=True)
            del H, gs, sv
        del g, gt
        torch.cuda.empty_cache()


def main():
    dev = torch.device("cuda:0")
    if "--train" in sys.argv:
        return train_variants(dev)
    reps = 7
    for n in (250_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
        g = synth.random_regular_graph(n, 50, 1234, dev)
        gen = torch.Generator(device=dev).manual_seed(1)
        H = torch.randn((n, 64), device=dev, generator=gen)
        f1 = torch.randn((n, 8), device=dev, generator=gen)
        a2 = torch.randn((8, 8), device=dev, generator=gen) * 0.3
        b2 = torch.zeros(8, device=dev)
        c = torch.zeros(64, device=dev)
        out = torch.empty((n, 64), device=dev)
        for _ in range(2):
            ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out)
        times = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        times.sort()
        ms = times[len(times) // 2]
        alg = g.nnz * 292 + n * 296
        print(json.dumps({"kernel": "node_attn_fwd (eval)", "N": n, "E": g.nnz,
                          "H_table_MB": round(n * 256 / 1e6, 1), "ms": round(ms, 4),
                          "algorithmic_GBs": round(alg / ms / 1e6, 1),
                          "frac_of_8TBs": round(alg / ms / 1e6 / 8000, 4)}), flush=True)
        del g, H, f1, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
