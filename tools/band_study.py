#!/usr/bin/env python
"""Band study on the CPU oracle (oracle/poa_oracle.cpp, Engine::band_study): how often would an exact banded DP with a
window of WB columns per row be certified, on the configurations of BASELINE.json?  Test infrastructure; the kernel's
banded DP (racon_amd/csrc/poa_kernel2.hpp) follows the same window policy and certificate."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_lib
from racon_amd.synth import simulate_windows, config_windows

def study(batch, scores, wb, g, tag):
    lib = oracle_lib.lib()
    lib.rcn_oracle_band_study.argtypes = [C.c_int, C.c_int]
    lib.rcn_oracle_band_study.restype = None
    lib.rcn_oracle_band_stats.argtypes = [C.POINTER(C.c_uint64)]
    lib.rcn_oracle_band_stats.restype = None
    lib.rcn_oracle_band_study(wb, g)
    t = time.time()
    oracle_lib.consensus(batch, *scores, True, 0)
    lib.rcn_oracle_band_study(0, 0)
    out = (C.c_uint64 * 10)()
    lib.rcn_oracle_band_stats(out)
    n, banded, ex, ch, same, bad, wmax, wsum, rows, shifts = [int(v) for v in out]
    print(f"{tag}: wb {wb} g {g}: alignments {n}, banded {banded}, exact-cert ok {ex} ({ex/max(1,banded):.4f}), cheap-cert ok {ch} ({ch/max(1,banded):.4f}), "
          f"same result {same} ({same/max(1,banded):.4f}), cert-ok-but-different {bad}, max alive width {wmax}, mean alive width {wsum/max(1,rows):.1f}, "
          f"shifts/alignment {shifts/max(1,banded):.1f}  [{time.time()-t:.1f}s]", flush=True)

if __name__ == "__main__":
    b = simulate_windows(100_000, 500, 30.0, 10000, seed=20260921)
    for wb, g in ((256, 16), (192, 16), (128, 8), (128, 16)):
        study(b, (3, -5, -4), wb, g, "cfg2-like 200 windows")
    study(b, (5, -4, -8), 256, 16, "cfg2-like, scores 5/-4/-8")
    study(b, (1, -1, -1), 256, 16, "cfg2-like, scores 1/-1/-1")
    b = simulate_windows(50_000, 1000, 30.0, 10000, seed=20260925)
    for wb in (256, 512):
        study(b, (3, -5, -4), wb, 16, "w1000 50 windows")
