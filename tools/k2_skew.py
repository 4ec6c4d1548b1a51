#!/usr/bin/env python3
"""K2 on the power-law graph of SURVEY.md 8d (one meta-path of syn-1m-skew): the three kernels timed alone for a few
launch shapes -- degree bins on / off, and where rows start to be cut into chunks (SPLIT_DEG / SPLIT_CHUNK).
One JSON line per setting (median of 7 launches, HIP events)."""
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
    wl = kv.get("workload", "syn-1m-skew")
    cfg = synth.CONFIGS[wl]
    n = cfg["n"]
    g = synth.make_graph(cfg["graphs"][0], n, 1234, dev)
    gt = g.transpose()
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=gen)
    tdt = torch.bfloat16 if kv.get("bf16") == "1" else torch.float32
    a1, a2, b1, b2 = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1
    c = rnd(64) * 0.1
    X, W = rnd(n, 64), torch.eye(64, device=dev)
    H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3, table_dtype=tdt)
    out = torch.empty((n, 64), device=dev)
    dOut = rnd(n, 64)
    deg = g.degrees()
    print(json.dumps({"workload": wl, "E": g.nnz, "rows_below_16": int((deg < 16).sum()), "rows_above_8192": int((deg > 8192).sum()),
                      "max_degree": int(deg.max()), "transposed_max_degree": int(gt.degrees().max())}), flush=True)
    settings = [(False, 8192, 4096), (True, 8192, 4096), (True, 2048, 1024), (True, 1024, 512), (True, 512, 512), (True, 512, 256),
                (True, 256, 256)]
    if "settings" in kv:
        settings = [tuple(int(v) for v in s.split(":")) for s in kv["settings"].split(",")]
        settings = [(bool(b), d, ch) for b, d, ch in settings]
    for binned, sd, ch in settings:
        ops.BINNED, ops.SPLIT_DEG, ops.SPLIT_CHUNK = binned, sd, ch
        t_e = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out))
        t_t = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=3))
        _, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=3)
        o_, lse, aggp, tsum = sv
        gs, df1, _ = ops.node_attn_bwd_rows(dOut, o_, aggp, tsum, f1, lse, c, table_dtype=tdt)
        t_b = timeit(lambda: ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3))
        sp = g.row_split(sd, ch)
        print(json.dumps({"binned": binned, "split_deg": sd, "chunk": ch, "n_long": sp["n_long"] if sp else 0,
                          "n_chunks": sp["n_chunks"] if sp else 0, "eval_ms": round(t_e, 4), "train_ms": round(t_t, 4),
                          "bwd_cols_ms": round(t_b, 4)}), flush=True)


if __name__ == "__main__":
    main()
