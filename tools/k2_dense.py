#!/usr/bin/env python3
"""K2 on small dense graphs: the matrix-pipe form (bit mask + fp32 MFMA tiles, exp-free scores) against the lean CSR
kernels and the classic gather kernels -- times per launch and the largest deviation of every output.  One JSON line
per (n, density)."""
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
    cases = [(4057, 12924399), (4057, 5000495), (3025, 2210761), (4057, 0.12), (4057, 0.05), (3025, 29281)]
    if "n" in kv:
        cases = [(int(kv["n"]), int(d) if float(d) > 1 else float(d)) for d in kv.get("dens", "0.5").split(",")]
    a1, a2, b1, b2 = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1
    c = rnd(64) * 0.1
    for n, spec in cases:
        g = synth.bernoulli_graph(n, None, 7, dev, nnz=spec) if spec > 1 else synth.bernoulli_graph(n, spec, 7, dev)
        gt = g.transpose()
        X, W = rnd(n, 64), torch.eye(64, device=dev)
        H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3)
        dOut = rnd(n, 64)
        res = {}
        for mode in ("gather", "lean", "dense"):
            ops.LEAN, ops.DENSE = mode != "gather", mode == "dense"
            used = {"gather": True, "lean": ops._use_lean(g, H), "dense": ops._use_dense(g, H, 8, 8)}[mode]
            out_e, _ = ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2)
            out_e = out_e.clone()
            t_e = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, f2=f2))
            out_t, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.6, fts_drop=0.6, seed=3, f2=f2)
            t_t = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, train=True, coef_drop=0.6, fts_drop=0.6, seed=3, f2=f2))
            gs, df1, _ = ops.node_attn_bwd_rows(dOut, sv[0], sv[2], sv[3], f1, sv[1], c)
            dH, df2 = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3)
            t_b = timeit(lambda: ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3))
            res[mode] = dict(used=used, eval_ms=t_e, train_ms=t_t, bwd_ms=t_b, out_e=out_e, out_t=sv[0].clone(),
                             lse=sv[1], aggp=sv[2], tsum=sv[3], dH=dH, df2=df2)
        a = res["gather"]
        line = {"n": n, "edges": g.nnz, "density": round(g.nnz / n / n, 4),
                "gather_ms": {k: round(a[k], 4) for k in ("eval_ms", "train_ms", "bwd_ms")}}
        for mode in ("lean", "dense"):
            b = res[mode]
            diff = {k: float((a[k] - b[k]).abs().max()) for k in ("out_e", "out_t", "lse", "aggp", "tsum", "dH", "df2")}
            line[mode] = {"taken": b["used"], "eval_ms": round(b["eval_ms"], 4), "train_ms": round(b["train_ms"], 4),
                          "bwd_ms": round(b["bwd_ms"], 4), "max_abs_diff": {k: float(f"{v:.3g}") for k, v in diff.items()}}
        print(json.dumps(line), flush=True)
    ops.LEAN = ops.DENSE = True


if __name__ == "__main__":
    main()
