#!/usr/bin/env python
"""Times rcn_engine_build_windows (windows built in HBM, reference src/polisher.cpp:388-461) on a cfg2-shaped input
(1 Mbp contig, 30x of 10 kb ONT-like reads, -w 500) and on the committed reference fixture; prints one JSON line.
The gather is pure HBM traffic: bases + qualities read once and written once (4 bytes per packed base)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--contig", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--oracle", action="store_true", help="also time oracle/window_layout.py (pure Python, slow)")
args = ap.parse_args()

from racon_amd.engine import HipEngine  # noqa: E402
from racon_amd.synth import simulate_layout  # noqa: E402

t0 = time.time()
r, o, wt, al = simulate_layout(contig_lens=(args.contig,), coverage=30.0, read_len=10000, seed=20260921, with_cigars=True)
t_sim = time.time() - t0
eng = HipEngine()
best = None
for _ in range(args.reps):
    t0 = time.time()
    eng.build_windows(r, o, 500, 10.0, wt)
    wall = time.time() - t0
    st = eng.build_stats()
    st["wall_ms"] = wall * 1e3
    if best is None or st["kernel_ms"] < best["kernel_ms"]:
        best = st
b = eng.export_batch()
out = {"workload": f"cfg2-shaped layout: {args.contig} bp contig, {o.n_overlaps} overlaps, {best['n_pairs']} breaking-point pairs",
       "windows": b.n_windows, "layers": best["n_layers"], "packed_bases": int(len(b.bases)),
       "h2d_ms": round(best["h2d_ms"], 3), "device_ms": round(best["kernel_ms"], 3), "gather_ms": round(best["gather_ms"], 4),
       "gather_GBps": round(best["gather_bytes"] / (best["gather_ms"] * 1e-3) / 1e9, 1) if best["gather_ms"] > 0 else None,
       "wall_ms": round(best["wall_ms"], 2)}
if args.oracle:
    from oracle.window_layout import window_layout
    t0 = time.time()
    ref = window_layout(r, o, 500, 10.0, wt)
    out["python_oracle_s"] = round(time.time() - t0, 2)
    out["matches_oracle"] = bool((ref.bases == b.bases).all() and (ref.seq_off == b.seq_off).all() and (ref.seq_begin == b.seq_begin).all())
# the same with the CIGAR walk (reference src/overlap.cpp:226-292) on the device
bestc = None
for _ in range(args.reps):
    eng.build_windows_from_cigars(r, al, 500, 10.0, wt)
    st = eng.build_stats()
    if bestc is None or st["kernel_ms"] < bestc["kernel_ms"]:
        bestc = st
bc = eng.export_batch()
out["from_cigars"] = {"cigar_bytes": int(len(al.cigar)), "h2d_ms": round(bestc["h2d_ms"], 3), "device_ms": round(bestc["kernel_ms"], 3),
                      "same_batch": bool((bc.bases == b.bases).all() and (bc.seq_off == b.seq_off).all() and (bc.seq_begin == b.seq_begin).all()
                                         and (bc.seq_end == b.seq_end).all() and (bc.quals == b.quals).all())}
res = eng.run()
out["consensus_kernel_ms"] = round(eng.stats()["kernel_ms"], 2)
print(json.dumps(out))
