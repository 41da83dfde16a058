#!/usr/bin/env python3
"""How many cores does this host actually give a job?  N single-threaded processes each deflate the same 3 MB buffer `reps` times (zlib level 6, no I/O, no
shared state): perfect scaling keeps the wall time flat until N reaches the cores available to the container (cgroup quota, SMT siblings, other tenants).
Context for tools/host_scaling_probe.py: the 8-process host half needs ~19 cores' worth of PNG encoding per 0.8 s batch.

  python tools/host_cpu_probe.py > profiles/rNN_host_cpu_probe.txt
"""
import multiprocessing as mp
import os
import time
import zlib

import numpy as np


def work(reps):
    rng = np.random.default_rng(0)
    buf = (rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) >> 3).tobytes() * 4
    t = time.perf_counter()
    for _ in range(reps):
        zlib.compress(buf, 6)
    return time.perf_counter() - t


def main():
    print(f"# affinity cpus {len(os.sched_getaffinity(0))}, os.cpu_count {os.cpu_count()}, loadavg {open('/proc/loadavg').read().strip()}")
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f"# {f}: {open(f).read().strip()}")
    base = None
    for n in (1, 4, 8, 16, 32, 64, 128):
        with mp.Pool(n) as pool:
            t = time.perf_counter()
            per = pool.map(work, [6] * n)
            wall = time.perf_counter() - t
        base = base or max(per)
        print(f"{n:4d} processes: slowest {max(per) * 1e3:7.0f} ms, mean {sum(per) / n * 1e3:7.0f} ms, wall {wall * 1e3:7.0f} ms  -> effective cores {n * base / max(per):6.1f}")


if __name__ == "__main__":
    main()
