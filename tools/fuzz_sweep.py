#!/usr/bin/env python
"""tools/fuzz_sweep.py -- long fuzz of the small-window kernel (racon_amd/csrc/poa_small.hpp) against the CPU oracle.  GPU, untimed.

Per seed: 500 random tie-break stress windows (tests/test_gpu_fuzz.py: random_window, in the alphabets the kernel keeps plus one
window in eight with eight symbols) + directed windows, each built to make the kernel give a window back for ONE named reason
(rcn_run_stats.small_bail_why: graph capacity, ninth in-edge, far predecessor, ring beyond five symbols, int16 range, sink tie) --
through the engine (C ABI) for five score sets x trim on / off, and once more with RCN_FORCE_EXACT (spoa's own DFS order for every
consensus).  Every window of every run is compared with the oracle, byte for byte; `small_bail_why[8]` (internal inconsistency) must
stay zero.  Reference semantics under test: src/window.cpp:65-149 (the Subgraph branch :99-107 nearly every layer takes).

    python tools/fuzz_sweep.py --seeds 120 --out gpurun_out/r05a/fuzz_sweep.json

The oracle is the checker here (test infrastructure); nothing of it is measured or shipped.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RCN_EXPERIMENT", "1")
os.environ.setdefault("RCN_FORCE_SMALL", "1")     # engines are reused across seeds: the kernel stays on after a batch that sent many windows back

import numpy as np

SCORES = [(3, -5, -4), (5, -4, -8), (1, -1, -1), (2, -3, -2), (4, -6, -100)]
WHY = ["graph capacity", "ninth in-edge / seventeenth wide row", "far predecessor", "ring beyond five symbols", "int16 range", "layer too long",
       "sink tie beyond the id rule", "consensus scratch", "INTERNAL INCONSISTENCY"]


def qual(s: bytes, v: int) -> bytes:
    return bytes([33 + v]) * len(s)


def directed_windows(rng):
    """Windows that leave the kernel for a named reason (and come back through poa_window_kernel2 with the same bytes)."""
    wins = []
    acgt = list(b"ACGT")

    def backbone(n):
        return bytes(rng.choice(acgt, n).astype(np.uint8).tolist())

    # (2) a node with more than eight in-edges: deletions of 1..11 columns that all end in front of column 60
    bb = backbone(120)
    seqs = [(bb, qual(bb, 20), 0, 0)]
    for d in range(1, 12):
        lay = bb[:60 - d] + bb[60:]
        seqs.append((lay, qual(lay, 25), 0, 119))
    seqs += [(bb, qual(bb, 25), 0, 119)] * 2
    wins.append({"type": 1, "seqs": seqs})
    # (3) a predecessor more than sixteen rows back: one deletion of 30 columns, layers on both sides of it
    bb = backbone(150)
    lay = bb[:50] + bb[80:]
    seqs = [(bb, qual(bb, 20), 0, 0)] + [(bb, qual(bb, 25), 0, 149)] * 2 + [(lay, qual(lay, 30), 0, 149)] * 3 + [(bb[20:140], qual(bb[20:140], 25), 20, 139)] * 2
    wins.append({"type": 1, "seqs": seqs})
    # (4) a column with seven symbols: its aligned ring outgrows the kernel's five
    bb = backbone(120)
    lay = bb[10:110]
    seqs = [(bb, qual(bb, 20), 0, 0)] + [(lay, qual(lay, 25), 10, 109)] * 3
    for sym in b"CGTNRYK":
        alt = lay[:50] + bytes([sym]) + lay[51:]
        seqs.append((alt, qual(alt, 25), 10, 109))
    wins.append({"type": 0, "seqs": seqs})
    # (1) noisy reads on a 200-base window: the graph outgrows its LDS capacity (and (5) with gap -100: Z range)
    bb = backbone(200)
    seqs = [(bb, qual(bb, 15), 0, 0)]
    for _ in range(int(rng.integers(45, 70))):
        out = bytearray()
        for ch in bb:
            r = rng.random()
            if r < 0.06:
                continue
            out.append(int(rng.choice(acgt)) if r < 0.12 else ch)
            while rng.random() < 0.07:
                out.append(int(rng.choice(acgt)))
        s = bytes(out[:250]) or b"A"
        seqs.append((s, qual(s, int(rng.integers(5, 30))), 0, 199))
    wins.append({"type": 1, "seqs": seqs})
    # (7) sink ties: homopolymer backbone, layers that stop short of alternative ends
    bb = b"A" * 60
    seqs = [(bb, qual(bb, 10), 0, 0)]
    for k in range(12):
        s = b"A" * int(rng.integers(40, 70)) + (b"C" if k % 3 == 0 else b"G" if k % 3 == 1 else b"")
        seqs.append((s, qual(s, 10), 0, 59))
    wins.append({"type": int(rng.integers(0, 2)), "seqs": seqs})
    return wins


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=120)
    ap.add_argument("--first-seed", type=int, default=5000)
    ap.add_argument("--windows", type=int, default=500)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-small", action="store_true", help="RCN_NO_SMALL=1: every window through poa_window_kernel2 (and its retry tier) instead")
    a = ap.parse_args()
    if a.no_small:
        os.environ["RCN_NO_SMALL"] = "1"
        os.environ.pop("RCN_FORCE_SMALL", None)

    from racon_amd.batch import WindowBatch
    from racon_amd.engine import HipEngine
    from oracle import oracle_lib
    from test_gpu_fuzz import random_window

    t_start = time.time()
    tot = {"seeds": 0, "runs": 0, "windows_compared": 0, "mismatching_windows": 0, "n_small": 0, "n_small_bailed": 0, "n_retried": 0,
           "small_bail_why": [0] * 9, "first_mismatches": []}
    engines = {}

    def engine(scores, trim, exact):
        key = (scores, trim, exact)
        if key not in engines:
            if exact:
                os.environ["RCN_FORCE_EXACT"] = "1"
            else:
                os.environ.pop("RCN_FORCE_EXACT", None)
            engines[key] = HipEngine(*scores, trim)          # (switches are read once, when the engine is created)
            os.environ.pop("RCN_FORCE_EXACT", None)
        return engines[key]

    for s in range(a.first_seed, a.first_seed + a.seeds):
        rng = np.random.default_rng(s)
        wins = [random_window(rng, 5 * int(rng.integers(0, 50)) + (4 if rng.random() < 0.125 else int(rng.integers(0, 4)))) for _ in range(a.windows)]
        wins += directed_windows(rng)
        b = WindowBatch.from_windows(wins)
        for si, scores in enumerate(SCORES):
            for trim in (True, False):
                for exact in ((False, True) if (trim and si == (s % len(SCORES))) else (False,)):
                    ref = oracle_lib.consensus(b, *scores, trim, a.threads)
                    eng = engine(scores, trim, exact)
                    got = eng.consensus(b)
                    st = eng.stats()
                    bad = [k for k in range(b.n_windows) if got.consensus[k] != ref.consensus[k] or got.polished[k] != ref.polished[k] or got.chimeric[k] != ref.chimeric[k]]
                    tot["runs"] += 1
                    tot["windows_compared"] += b.n_windows
                    tot["mismatching_windows"] += len(bad)
                    tot["n_small"] += st["n_small"]; tot["n_small_bailed"] += st["n_small_bailed"]; tot["n_retried"] += st["n_retried"]
                    for k, v in enumerate(st["small_bail_why"]):
                        tot["small_bail_why"][k] += int(v)
                    for k in bad[:3]:
                        if len(tot["first_mismatches"]) < 20:
                            tot["first_mismatches"].append({"seed": s, "scores": scores, "trim": trim, "exact": exact, "window": k, "directed": k >= a.windows})
        tot["seeds"] += 1
        if (s - a.first_seed) % 10 == 9:
            print("seed %d: %d runs, %d windows compared, %d mismatches, small %d, bailed %d %s, %.0f s" %
                  (s, tot["runs"], tot["windows_compared"], tot["mismatching_windows"], tot["n_small"], tot["n_small_bailed"], tot["small_bail_why"], time.time() - t_start), flush=True)
    tot["seconds"] = round(time.time() - t_start, 1)
    tot["small_bail_why_named"] = {WHY[k]: tot["small_bail_why"][k] for k in range(9)}      # (index = reason - 1, as in rcn_run_stats)
    tot["ok_no_internal_inconsistency"] = tot["small_bail_why"][8] == 0
    tot["what"] = ("%d seeds x (%d random + 5 directed windows) x %d score sets x trim on/off, + RCN_FORCE_EXACT once per seed: HIP engine (small-window kernel "
                   "+ its retry tiers) vs oracle/poa_oracle.cpp, every window" % (a.seeds, a.windows, len(SCORES)))
    tot["kernel"] = "poa_window_kernel2 (RCN_NO_SMALL=1)" if a.no_small else "poa_window_kernel_small + retry tiers"
    tot["ok"] = tot["mismatching_windows"] == 0 and tot["small_bail_why"][8] == 0
    line = json.dumps(tot)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(json.dumps(tot, indent=1) + "\n")
    sys.exit(0 if tot["ok"] else 1)


if __name__ == "__main__":
    main()
