#!/usr/bin/env python3
"""K2 on small dense graphs (the reference's own data-set shapes): the lean kernels (HAN_FLAG_LEAN: scores read from
the table, shared dropout hash, one lane per head at 8 x 8) against the gather kernels on the same inputs -- time per
launch and the largest difference of the outputs.  One JSON line per graph.
`python tools/k2_small_dense.py [n=4057 dens=0.78,0.30,0.24]`"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import ops, synth  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402


def main():
    kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=gen)
    cases = [(4057, 0.785), (4057, 0.304), (3025, 0.2416), (3025, 0.05), (4057, 0.0007)]
    if "n" in kv:
        cases = [(int(kv["n"]), float(d)) for d in kv.get("dens", "0.5").split(",")]
    a1, a2, b1, b2 = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1
    c = rnd(64) * 0.1
    for n, dens in cases:
        g = synth.bernoulli_graph(n, dens, 7, dev)
        gt = g.transpose()
        X, W = rnd(n, 64), torch.eye(64, device=dev)
        H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3)
        dOut = rnd(n, 64)
        res = {}
        for mode in ("gather", "lean"):
            ops.LEAN = mode == "lean"
            used = ops._use_lean(g, H) if mode == "lean" else True
            out_e, _ = ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2)
            t_e = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2))
            out_t, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.6, fts_drop=0.6, seed=3, f2=f2)
            t_t = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.6, fts_drop=0.6, seed=3, f2=f2))
            gs, df1, _ = ops.node_attn_bwd_rows(dOut, sv[0], sv[2], sv[3], f1, sv[1], c)
            dH, df2 = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3)
            t_b = timeit(lambda: ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3))
            res[mode] = dict(used=used, eval_ms=t_e, train_ms=t_t, bwd_ms=t_b, out_e=out_e, out_t=out_t, pre=sv[0],
                              lse=sv[1], aggp=sv[2], tsum=sv[3], dH=dH, df2=df2)
        a = res["gather"]
        line = {"n": n, "density": dens, "edges": g.nnz,
                "gather_ms": {k: round(a[k], 4) for k in ("eval_ms", "train_ms", "bwd_ms")}}
        for mode in ("lean",):
            b = res[mode]
            diff = {k: float((a[k] - b[k]).abs().max()) for k in ("out_e", "out_t", "pre", "lse", "aggp", "tsum", "dH", "df2")}
            line[mode] = {"taken": b["used"], "eval_ms": round(b["eval_ms"], 4), "train_ms": round(b["train_ms"], 4),
                          "bwd_ms": round(b["bwd_ms"], 4),
                          "max_abs_diff": {k: float(f"{v:.3g}") for k, v in diff.items()}}
        print(json.dumps(line), flush=True)
    ops.LEAN = True


if __name__ == "__main__":
    main()
