#!/usr/bin/env python3
"""Host probe for bench.py's cpu_baseline: what the box's CPU share really is (cgroup quota, affinity) and how the
torch-CPU restatement scales with the thread count.  Usage: python tools/cpu_threads_probe.py [N] [threads ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    threads = [int(t) for t in sys.argv[2:]] or [8, 16, 32, 64]
    print("host_cores", bench.host_cores(), "cpu_count", os.cpu_count(), flush=True)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip(), flush=True)
    for t in threads:
        t0 = time.perf_counter()
        s, reps = bench._time_cpu_epochs("syn-1m", n, t, 2, 5.0)
        print(f"threads {t}: {s:.2f} s/epoch at N={n} ({reps} epochs, {time.perf_counter() - t0:.0f} s wall)", flush=True)
