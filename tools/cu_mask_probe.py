#!/usr/bin/env python3
"""How do the hot kernels scale with the number of CUs they may use?  Streams with a CU mask
(hipExtStreamCreateWithCUMask; c of every 32 CUs: bits i with i % 32 < c -- uniform over the XCDs whichever way the
mask bits are numbered), the SYN-1M kernels timed alone on each: K2 training forward, K2 backward gather, K1 training
forward, K1 dW.  One JSON line per (kernel, CUs)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import ops, synth  # noqa: E402


def masked_stream(hip, c):
    words = (ctypes.c_uint32 * 8)(*([(1 << c) - 1 if c < 32 else 0xFFFFFFFF] * 8))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def timeit(fn, stream, reps=5):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    hip = ctypes.CDLL("libamdhip64.so")
    n, f = 1_000_000, 256
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=gen)
    g = synth.random_regular_graph(n, 50, 1234, dev)
    gt = g.transpose()
    X = rnd(n, f)
    W = rnd(f, 64) * 0.05
    a1, a2, b1, b2, c = rnd(8, 8) * 0.3, rnd(8, 8) * 0.3, rnd(8) * 0.1, rnd(8) * 0.1, rnd(64) * 0.1
    H, f1, f2, keep = ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3, want_keep=True)
    out = torch.empty((n, 64), device=dev)
    _, sv = ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=3)
    pre, lse, aggp, tsum = sv
    dOut = rnd(n, 64)
    gs, df1, _ = ops.node_attn_bwd_rows(dOut, pre, aggp, tsum, f1, lse, c)
    dH, _ = ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3)
    torch.cuda.synchronize()
    kernels = {
        "k2 fwd train": lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out, train=True, coef_drop=0.6, fts_drop=0.6, seed=3),
        "k2 fwd eval": lambda: ops.node_attn_fwd(g, H, f1, a2, b2, c, out=out),
        "k2 bwd cols": lambda: ops.node_attn_bwd_cols(gt, gs, H, f2, df1, a1, a2, coef_drop=0.6, fts_drop=0.6, seed=3),
        "k1 fwd train": lambda: ops.project_fwd(X, W, a1, a2, b1, b2, in_drop=0.6, fts_drop=0.6, seed=3, want_keep=True),
        "k1 dW": lambda: ops.project_bwd(X, dH, 8, 8, in_drop=0.6, seed=3, keep=keep),
    }
    for cus in (32, 28, 24, 20, 16, 12, 8):
        st = masked_stream(hip, cus)
        for name, fn in kernels.items():
            print(json.dumps({"kernel": name, "cus": cus * 8, "ms": round(timeit(fn, st), 4)}), flush=True)


if __name__ == "__main__":
    main()
