#!/usr/bin/env python
"""One window of tools/fuzz_sweep.py again, through every kernel path (GPU): which path disagrees with the oracle, and how.
usage: python tools/exp/fuzz_repro.py <seed> <window> <m,x,g> [trim]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["RCN_EXPERIMENT"] = "1"


def child(seed, k, scores, trim, alone):
    import numpy as np
    from racon_amd.batch import WindowBatch
    from racon_amd.engine import HipEngine
    from oracle import oracle_lib
    import fuzz_sweep as F
    from test_gpu_fuzz import random_window
    rng = np.random.default_rng(seed)
    wins = [random_window(rng, 5 * int(rng.integers(0, 50)) + (4 if rng.random() < 0.125 else int(rng.integers(0, 4)))) for _ in range(500)]
    wins += F.directed_windows(rng)
    if alone:
        wins = [wins[k]] * 70          # (a batch of at least 64 windows takes the streamed path; fewer the plain one)
        k = 0
    b = WindowBatch.from_windows(wins)
    ref = oracle_lib.consensus(b, *scores, trim, 2)
    eng = HipEngine(*scores, trim)
    got = eng.consensus(b)
    st = eng.stats()
    bad = [i for i in range(b.n_windows) if got.consensus[i] != ref.consensus[i]]
    print("   n_small %d bailed %d why %s retried %d | mismatching windows %s" % (st["n_small"], st["n_small_bailed"], st["small_bail_why"], st["n_retried"], bad[:8]))
    if bad:
        i = bad[0]
        print("   oracle:", ref.consensus[i].decode(errors="replace"))
        print("   hip   :", got.consensus[i].decode(errors="replace"))
    w = wins[k]
    if os.environ.get("REPRO_PRINT"):
        print("   window type %d, %d sequences" % (w["type"], len(w["seqs"])))
        for s, q, b0, e0 in w["seqs"]:
            print("     [%d, %d] %s %s" % (b0, e0, s.decode(), "(no quality)" if q is None else q.decode(errors="replace")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        seed, k = int(sys.argv[2]), int(sys.argv[3]); scores = tuple(int(v) for v in sys.argv[4].split(",")); trim = sys.argv[5] == "1"; alone = sys.argv[6] == "1"
        child(seed, k, scores, trim, alone)
        sys.exit(0)
    seed, k = int(sys.argv[1]), int(sys.argv[2]); scores = sys.argv[3]; trim = sys.argv[4] if len(sys.argv) > 4 else "1"
    first = True
    for alone in ("0", "1"):
        for label, env in (("default", {}), ("RCN_NO_SMALL", {"RCN_NO_SMALL": "1"}), ("RCN_FORCE_EXACT", {"RCN_FORCE_EXACT": "1"}),
                           ("RCN_NO_SMALL + RCN_FORCE_EXACT", {"RCN_NO_SMALL": "1", "RCN_FORCE_EXACT": "1"}), ("RCN_FORCE_TIE=2", {"RCN_FORCE_TIE": "2"}),
                           ("RCN_FORCE_TIE=3", {"RCN_FORCE_TIE": "3"}), ("RCN_NO_SMALL + RCN_FORCE_TIE=3", {"RCN_NO_SMALL": "1", "RCN_FORCE_TIE": "3"}),
                           ("RCN_NO_BAND + RCN_NO_SMALL", {"RCN_NO_SMALL": "1", "RCN_NO_BAND": "1"}), ("RCN_WIDE_ONLY (int32 kernel)", {"RCN_WIDE_ONLY": "1"})):
            print("== %s, %s" % (label, "the window alone (x 70)" if alone == "1" else "in its batch"), flush=True)
            e = dict(os.environ, **env)
            if first:
                e["REPRO_PRINT"] = "1"; first = False
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(seed), str(k), scores, trim, alone], env=e)
