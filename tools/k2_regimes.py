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


def main():
    dev = torch.device("cuda:0")
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
