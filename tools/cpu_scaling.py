import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from racon_amd.synth import simulate_windows
from oracle import oracle_lib
b = simulate_windows(256000, 500, 30, 10000, seed=20260921)
print("windows", b.n_windows)
for t in (1, 8, 32, 64, 128, 256):
    n = min(b.n_windows, max(8, 4 * t))
    s = b.select(range(n))
    t0 = time.perf_counter(); oracle_lib.consensus(s, 3, -5, -4, True, t); dt = time.perf_counter() - t0
    print("threads %d windows %d: %.2fs  %.1f windows/s  (%.2f/thread)" % (t, n, dt, n / dt, n / dt / t), flush=True)
