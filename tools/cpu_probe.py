#!/usr/bin/env python
"""How many host cores does this process really get?  N threads of a fixed integer spin (C, no memory traffic, the GIL
released: it is the oracle library's edit-distance routine on a fixed pair) at N = 8 .. all logical CPUs: aggregate
throughput that stops growing at N = k says the lease is worth k cores whatever nproc prints.  Read next to bench.py's
cpu_baseline thread sweep (does the oracle port stop scaling where the BOX stops, or earlier?)."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_lib

rng = np.random.default_rng(1)
a = bytes(rng.integers(65, 69, 6000).astype(np.uint8)); b = bytes(rng.integers(65, 69, 6000).astype(np.uint8))
lib = oracle_lib.lib()


def spin(n, out, i):
    for _ in range(n):
        out[i] += 1 if lib.rcn_oracle_edit_distance(a, len(a), b, len(b)) else 0


ncpu = len(os.sched_getaffinity(0))
res = {}
for n in sorted({1, 8, 16, 32, 48, 64, 96, 128, 192, ncpu}):
    if n > ncpu:
        continue
    out = [0] * n
    th = [threading.Thread(target=spin, args=(40, out, i)) for i in range(n)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    res[n] = round(sum(out) / dt, 1)
base = res[min(res)] / min(res)
print(json.dumps({"logical_cpus": ncpu, "calls_per_s_by_threads": res, "speedup_over_one_thread_estimate": {k: round(v / base, 1) for k, v in res.items()}}))
