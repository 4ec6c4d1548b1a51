#!/usr/bin/env python3
"""HANTrainer(use_graph=True, overlap_eval=True) against the plain captured epoch on a bench workload: the first
epoch whose training pair or parameters differ (none is expected), and two overlapped runs against each other."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from han_amd import rng as hrng, synth  # noqa: E402
from han_amd.gat import HeteGAT_multi  # noqa: E402
from han_amd.trainer import HANTrainer  # noqa: E402


FORM = "branch"


def run(wl, overlap, epochs, dev, every=1):
    torch.manual_seed(0)
    hrng.manual_seed(5)
    model = HeteGAT_multi().build(wl["p"], wl["f"], wl["c"], (8,), (8, 1), 128, device=dev)
    tr = HANTrainer(model, [wl["x"]] * wl["p"], wl["graphs"], wl["labels"], wl["train_mask"], wl["val_mask"],
                    attn_drop=0.6, ffd_drop=0.6, use_graph=True, overlap_eval=FORM if overlap else False)
    hist, flats = [], []
    if every == 0:                      # back-to-back replays, as bench.py issues them: no host sync in between
        for k in range(epochs):
            out = tr.epoch()
        torch.cuda.synchronize()
        return [[float(v) for v in out]], [model.flat.detach().clone()]
    for k in range(epochs):
        out = [float(v) for v in tr.epoch()]
        hist.append(out)
        if k % every == 0:
            flats.append(model.flat.detach().clone())
    return hist, flats


def first_diff(a, b, what):
    for k, (u, v) in enumerate(zip(a, b)):
        same = torch.equal(u, v) if isinstance(u, torch.Tensor) else u == v
        if not same:
            return {"what": what, "first_difference_at": k}
    return {"what": what, "first_difference_at": None}


def main():
    global FORM
    kv = dict(a.split("=") for a in sys.argv[1:])
    FORM = kv.get("form", "branch")
    dev = torch.device("cuda:0")
    wl = synth.make_workload(kv.get("workload", "acm-like"), device=dev)
    n = int(kv.get("epochs", 200))
    every = int(kv.get("every", 1))
    plain, fp = run(wl, False, n, dev, every)
    if kv.get("quick") == "1":
        for rep in range(int(kv.get("reps", 3))):
            o1, f1 = run(wl, True, n, dev, every)
            print(json.dumps(dict(first_diff(fp, f1, "parameters: plain vs overlapped"), rep=rep, every=every,
                                  form=FORM)))
        return
    plain2, fp2 = run(wl, False, n, dev)
    o1, f1 = run(wl, True, n, dev)
    o2, f2 = run(wl, True, n, dev)
    print(json.dumps(first_diff(fp, fp2, "parameters: plain vs plain")))
    print(json.dumps(first_diff(fp, f1, "parameters: plain vs overlapped")))
    print(json.dumps(first_diff(f1, f2, "parameters: overlapped vs overlapped")))
    print(json.dumps(first_diff([h[:2] for h in plain], [h[:2] for h in o1], "training pair: plain vs overlapped")))
    print(json.dumps(first_diff([h[2:] for h in plain[:-1]], [h[2:] for h in o1[1:]], "validation pair (shifted), exact")))
    print(json.dumps({"plain_last": plain[-1], "overlapped_last": o1[-1], "overlapped2_last": o2[-1]}))


if __name__ == "__main__":
    main()
