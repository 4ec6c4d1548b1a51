#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles/k2_traffic.json:
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in KB; gfx950 tallies
128-B fetch requests as 64 B -- checked against the K1 eval launch, which must read X = N*F*4 B)."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"k2_fwd_eval": "node_attn_fwd_kernel<8, false, 1,", "k2_fwd_train": "node_attn_fwd_kernel<8, true, 1,",
           "k2_bwd_cols": "node_attn_bwd_cols_kernel<8, 1,", "k1_fwd_eval_calibration": "project_fwd_kernel<8, false,"}


def main():
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_traffic", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            for tag, sub in KERNELS.items():
                if sub in r["Kernel_Name"]:
                    vals[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
    detail = {}
    for tag, d in vals.items():
        fetch = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        write = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        detail[tag] = {"FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
                       "launches": [len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])],
                       "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    import datetime
    import hashlib
    h = hashlib.sha256()
    for f in ("han_amd/csrc/node_attn.hip", "han_amd/csrc/han_common.h"):     # == bench.py _src_sha()
        h.update(open(os.path.join(ROOT, f), "rb").read())
    out = {"workload": "syn-1m", "n_gpus": 1, "round": 2, "date": datetime.date.today().isoformat(),
           "kernel_src_sha": h.hexdigest()[:16],
           "kernel": "node_attn_fwd_kernel<8,false,1,4> (K2 forward, eval)",
           "hbm_bytes_per_launch": detail["k2_fwd_eval"]["hbm_bytes_per_launch"],
           "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 B; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                      "128-B requests at 64 B); calibration in the same run: project_fwd_kernel<8,false> reads "
                      "X = 1.024 GB (+W from L2)",
           "note": "fabric-side counter: Infinity-Cache hits are included, so this is L2-miss traffic, an upper bound "
                   "on HBM bytes; at SYN-1M the 256 MB H table sits in the 256 MiB Infinity Cache",
           "detail": detail,
           "command": "bash tools/pmc_traffic.sh && python3 tools/pmc_traffic.py"}
    old = os.path.join(ROOT, "profiles", "k2_traffic.json")
    if os.path.exists(old):
        try:
            out["history"] = json.load(open(old)).get("history", {})
        except Exception:
            pass
    json.dump(out, open(old, "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in detail.items()}))


if __name__ == "__main__":
    main()
