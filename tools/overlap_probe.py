#!/usr/bin/env python3
"""Does the MFMA/VALU-bound projection (K1) hide under the bandwidth-bound node attention
(K2) when they run on two HIP streams?  SYN-1M shape, one meta-path each."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = synth.make_workload("syn-1m", device=dev)
    g = wl["graphs"][0]
    gt = g.transpose()
    n, f = wl["n"], wl["f"]
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=gen)
    X, W = wl["x"], rnd(f, 64) * 0.1
    a1, a2, b1, b2, c = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1, rnd(64) * 0.1
    H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3)
    out = torch.empty((n, 64), device=dev)
    dH = rnd(n, 64)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def k1f(): ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=5)
    def k1e(): ops.project_fwd(X, W, a1, a2, b1, b2)
    def k1b(): ops.project_bwd(X, dH, 8, 8, in_drop=0.6, seed=5)
    def k2t(): ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=7)
    def k2e(): ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out)
    _, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=7)
    gs, df1, _ = ops.node_attn_bwd_rows(dH, sv[0], sv[2], sv[3], f1, sv[1], c)
    def k2b(): ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=7)

    def wall(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    def both(a, b, a_first=True):
        def run():
            side.wait_stream(main_s)
            if a_first:
                with torch.cuda.stream(side):
                    a()
                b()
            else:
                b()
                with torch.cuda.stream(side):
                    a()
            main_s.wait_stream(side)
        return run

    for na, a in (("k1_train_fwd", k1f), ("k1_eval_fwd", k1e), ("k1_bwd", k1b)):
        for nb, b in (("k2_train_fwd", k2t), ("k2_eval_fwd", k2e), ("k2_bwd_cols", k2b)):
            ta, tb = wall(a), wall(b)
            t1, t2 = wall(both(a, b, True)), wall(both(a, b, False))
            print(json.dumps({"side": na, "main": nb, "alone_ms": [round(ta, 3), round(tb, 3)],
                              "sum_ms": round(ta + tb, 3), "concurrent_side_first_ms": round(t1, 3),
                              "concurrent_main_first_ms": round(t2, 3)}), flush=True)


if __name__ == "__main__":
    main()
