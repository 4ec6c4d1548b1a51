#!/usr/bin/env python3
"""Stand-alone timings of K1 / K3 kernels at the SYN-1M shape (N = 1M, P = 4,
F = 256): one JSON line per kernel (median of `reps`, HIP events on the launch
stream).  Used while tuning; `bench.py` is the headline measurement."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import ops  # noqa: E402


def timeit(fn, reps=7):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    which = [a for a in sys.argv[1:] if "=" not in a] or ["k1", "k3"]
    kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)      # e.g. n=3025 f=1870 (ACM shape)
    dev = torch.device("cuda:0")
    n, p, f = int(kv.get("n", 1_000_000)), int(kv.get("p", 4)), int(kv.get("f", 256))
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=gen)
    if "k3" in which:
        M = rnd(n, p, 64)
        w, b, u = rnd(64, 128) * 0.1, rnd(128) * 0.1, rnd(128) * 0.1
        dZ = rnd(n, 64)
        Z, beta = ops.sem_attn_fwd(M, w, b, u)
        t = timeit(lambda: ops.sem_attn_fwd(M, w, b, u))
        fl = 2.0 * n * p * 64 * 128
        print(json.dumps({"kernel": "sem_attn_fwd", "ms": round(t, 4), "TFLOPs": round(fl / t / 1e9, 1)}))
        # k3forms=1: also the measurement-only forms (HAN_FLAG_K3_G3_F32 = 32, HAN_FLAG_K3_PAIRS = 16), same process
        forms = (("", 0), (" [G3 on the fp32 pipe]", 32), (" [two waves share a tile]", 16)) * 3 \
            if kv.get("k3forms") == "1" else (("", 0),)
        for tag, fl_ in forms:
            t = timeit(lambda: ops.sem_attn_bwd(M, w, b, u, beta, dZ, flags=fl_))
            print(json.dumps({"kernel": "sem_attn_bwd" + tag, "ms": round(t, 4), "TFLOPs": round(3 * fl / t / 1e9, 1)}))
        del M, dZ, Z, beta
    if "k2" in which:
        from han_amd import synth
        g = synth.random_regular_graph(n, 50, 1234, dev)
        gt = g.transpose()
        a1, a2, b1, b2 = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1
        c = rnd(64) * 0.1
        X = rnd(n, 64)
        W = torch.eye(64, device=dev)
        for tdt in (torch.float32, torch.bfloat16):
            H, f1, f2 = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3, table_dtype=tdt)
            out = torch.empty((n, 64), device=dev)
            tag = "f32" if tdt == torch.float32 else "bf16"
            t = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out))
            print(json.dumps({"kernel": f"k2 fwd eval {tag}", "ms": round(t, 4)}))
            for cd, fd in ((0.0, 0.0), (0.6, 0.0), (0.0, 0.6), (0.6, 0.6)):
                t = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=cd,
                                                     fts_drop=fd, seed=3))
                print(json.dumps({"kernel": f"k2 fwd train {tag} coef_drop={cd} fts_drop={fd}", "ms": round(t, 4)}))
            if kv.get("shared") == "1":       # A/B in one process: HAN_FLAG_K2_SHARED_HASH off / on, three rounds
                for rnd_ in range(3):
                    for sh in (False, True):
                        ops.K2_SHARED_HASH = sh
                        t = timeit(lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6,
                                                             fts_drop=0.6, seed=3))
                        print(json.dumps({"kernel": f"k2 fwd train {tag} shared_hash={int(sh)}", "n": n, "ms": round(t, 4)}))
                ops.K2_SHARED_HASH = False
            _, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=3)
            pre, lse, aggp, tsum = sv
            dOut = rnd(n, 64)
            gs, df1, dc = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, f1, lse, c, table_dtype=tdt)
            t = timeit(lambda: ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, f1, lse, c, table_dtype=tdt))
            print(json.dumps({"kernel": f"k2 bwd rows {tag}", "ms": round(t, 4)}))
            for cd in (0.0, 0.6):
                t = timeit(lambda: ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=cd,
                                                          fts_drop=0.6, seed=3))
                print(json.dumps({"kernel": f"k2 bwd cols {tag} coef_drop={cd}", "ms": round(t, 4)}))
        del g, gt, X, H
    if "k1" in which:
        X = rnd(n, f)
        if kv.get("xbf") == "1":          # bf16 features (configs[4])
            X = X.to(torch.bfloat16)
        W = rnd(f, 64) * 0.1
        a1, a2, b1, b2 = rnd(8, 8), rnd(8, 8), rnd(8), rnd(8)
        dH = rnd(n, 64)
        fl = 2.0 * n * f * 64
        fl_ = int(kv.get("flags", 0))
        for drop in (0.0, 0.6):
            t = timeit(lambda: ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=drop, fts_drop=drop, seed=5, flags=fl_))
            print(json.dumps({"kernel": f"project_fwd drop={drop} flags={fl_}", "ms": round(t, 4),
                              "TFLOPs": round(fl / t / 1e9, 1), "X_GBs": round(n * f * 4 / t / 1e6, 1)}))
            t = timeit(lambda: ops.project_bwd(X, dH, 8, 8, in_drop=drop, seed=5))
            print(json.dumps({"kernel": f"project_bwd drop={drop} (draws regenerated)", "ms": round(t, 4),
                              "TFLOPs": round(fl / t / 1e9, 1)}))
        # round 3: all P meta-paths of the shared X in one fused eval launch
        Wp = rnd(p, f, 64) * 0.1
        ap1, ap2, bp1, bp2 = rnd(p, 8, 8), rnd(p, 8, 8), rnd(p, 8), rnd(p, 8)
        for mf in (0, 32):
            t = timeit(lambda: ops.project_fwd_multi(X, Wp, ap1, ap2, bp1, bp2, flags=mf))
            print(json.dumps({"kernel": f"project_fwd_multi eval P={p} flags={mf}", "ms": round(t, 4),
                              "TFLOPs": round(p * fl / t / 1e9, 1)}))
        t = timeit(lambda: [ops.project_fwd(X, Wp[i], ap1[i], ap2[i], bp1[i], bp2[i]) for i in range(p)])
        print(json.dumps({"kernel": f"project_fwd eval x{p} (one launch per meta-path)", "ms": round(t, 4)}))
        # round 3: the forward writes the keep table, dW reads it (4x4x1 16-block MFMA kernel)
        t = timeit(lambda: ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=5, want_keep=True, flags=fl_))
        print(json.dumps({"kernel": f"project_fwd drop=0.6 + keep table flags={fl_}", "ms": round(t, 4)}))
        keep = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=5, want_keep=True)[3]
        if keep is not None:
            t = timeit(lambda: ops.project_bwd(X, dH, 8, 8, in_drop=0.6, seed=5, keep=keep))
            print(json.dumps({"kernel": "project_bwd drop=0.6 from the keep table", "ms": round(t, 4),
                              "TFLOPs": round(fl / t / 1e9, 1)}))


if __name__ == "__main__":
    main()
