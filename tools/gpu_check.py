"""Quick GPU parity + timing probe (dev tool): HIP engine vs oracle on synthetic windows."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from racon_amd.engine import HipEngine
from racon_amd.synth import simulate_windows
from oracle import oracle_lib

ap = argparse.ArgumentParser()
ap.add_argument("--contig", type=int, default=50000)
ap.add_argument("--w", type=int, default=500)
ap.add_argument("--cov", type=float, default=30)
ap.add_argument("--rl", type=int, default=10000)
ap.add_argument("--short", action="store_true")
ap.add_argument("--scores", default="3,-5,-4")
ap.add_argument("--slots", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
m, x, g = [int(v) for v in a.scores.split(",")]
kw = dict(sub=0.003, ins=0.0005, dele=0.0005, phred_mean=30, phred_sd=0) if a.short else {}
b = simulate_windows(a.contig, a.w, a.cov, a.rl, seed=a.seed, **kw)
print("windows", b.n_windows, "seqs", b.n_seqs, "bases", b.bases.size, flush=True)
eng = HipEngine(m, x, g, True, max_slots=a.slots)
t = time.time(); eng.upload(b); print("upload %.3fs" % (time.time() - t), flush=True)
for rep in range(a.reps):
    t = time.time(); r = eng.run(); dt = time.time() - t
    st = eng.stats()
    print("run %d: wall %.3fs kernel %.1f ms  %.1f windows/s  GCUPS %.2f  retried %d" % (
        rep, dt, st["kernel_ms"], b.n_windows / (st["kernel_ms"] / 1e3), st["dp_cells"] / st["kernel_ms"] / 1e6, st["n_retried"]), flush=True)
print(json.dumps(st))
names = ["sub", "desc", "dp", "traceback", "add", "toposort", "consensus", "other"]
tot = sum(st["phase_clocks"]) or 1
print("phases:", ", ".join("%s %.1f%%" % (n, 100.0 * c / tot) for n, c in zip(names, st["phase_clocks"])))
if not a.no_oracle:
    t = time.time(); o, cells, cx = oracle_lib.consensus(b, m, x, g, True, 0, with_stats=True); dt = time.time() - t
    print("oracle %.2fs (%d threads) %.1f windows/s; cells %d (device %d)" % (dt, os.cpu_count(), b.n_windows / dt, cells.sum(), st["dp_cells"]))
    bad = [i for i in range(b.n_windows) if o.consensus[i] != r.consensus[i] or o.polished[i] != r.polished[i] or o.chimeric[i] != r.chimeric[i]]
    print("MISMATCHES", len(bad), bad[:10])
    sys.exit(1 if bad else 0)
