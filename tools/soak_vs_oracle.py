#!/usr/bin/env python
"""Soak: synthetic windows of many shapes through the HIP engine against the CPU oracle (AVX2 variant) until --seconds are used up.
Every round draws a workload -- window length, coverage, read length, error rates, backbone errors, score set, trim -- generates ~`--mbp`
Mbp of it in worker processes, polishes it on the GPU and on the CPU and compares every window; a window that differs is written to
--out as a one-window .npz.  Prints one JSON line.  (What found this round's kernel bug was a check of this kind at size, tools/cfg5_full_check.py:
real jobs hold combinations the hand-made fuzz shapes do not.)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=600)
ap.add_argument("--mbp", type=float, default=6.0)
ap.add_argument("--workers", type=int, default=16)
ap.add_argument("--seed", type=int, default=20260930)
ap.add_argument("--out", default="gpurun_out/soak")
ap.add_argument("--fragment", action="store_true", help="fragment-correction windows (racon -f: the reads are the targets, dual overlaps) instead: cfg5's shape with drawn read lengths, depths and error rates")
ap.add_argument("--lowcomplexity", action="store_true", help="contigs with homopolymer runs, short tandem repeats and two-letter stretches (many co-optimal alignments: ties at every level), "
                "some with N bases, some at 80-150x (ninth in-edges, full rings), higher error rates")
a = ap.parse_args()


def piece(job):
    from racon_amd.synth import simulate_windows, simulate_fragment_windows
    n, kw = job
    if "n_reads" in kw:
        return simulate_fragment_windows(n, **kw)
    return simulate_windows(n, **kw)


def main():
    import multiprocessing as mp
    from racon_amd.engine import HipEngine
    from oracle import oracle_lib
    os.makedirs(a.out, exist_ok=True)
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    rounds, total, differ, first = [], 0, 0, []
    engines = {}
    k = 0
    while time.time() - t0 < a.seconds:
        k += 1
        w = int(rng.choice([150, 200, 300, 500, 500, 500, 700, 1000]))
        short = w <= 300 and rng.random() < 0.6
        kw = dict(window_len=w, coverage=float(rng.choice([12, 20, 30, 30, 45, 60])), read_len=int(rng.choice([150, 250]) if short else rng.choice([3000, 10000, 20000])),
                  sub=float(rng.choice([0.003, 0.01]) if short else rng.choice([0.01, 0.03, 0.05])), ins=float(rng.choice([0.0005, 0.002]) if short else rng.choice([0.01, 0.03, 0.05])),
                  dele=float(rng.choice([0.0005, 0.002]) if short else rng.choice([0.01, 0.04, 0.06])), backbone_errors=float(rng.choice([0.0, 0.0, 0.01, 0.03])),
                  with_quality=bool(rng.random() < 0.8))
        if a.lowcomplexity:
            kw["low_complexity"] = float(rng.choice([0.3, 0.6, 0.9, 1.0])); kw["n_rate"] = float(rng.choice([0.0, 0.0, 0.002, 0.02]))
            if rng.random() < 0.3:
                kw["coverage"] = float(rng.choice([80, 100, 150]))
            if not short and rng.random() < 0.3:
                kw.update(sub=0.08, ins=0.06, dele=0.08)
        scores = [(3, -5, -4), (5, -4, -8), (1, -1, -1), (2, -3, -2)][int(rng.integers(0, 4))]
        trim = bool(rng.random() < 0.8)
        npieces = max(1, int(a.mbp * (30.0 / kw["coverage"]) * (0.5 if w >= 700 else 1.0)))
        jobs = [(1_000_000 if not short else 300_000, dict(kw, seed=int(rng.integers(1, 2**31)))) for _ in range(npieces)]
        if a.fragment:
            rl = int(rng.choice([3000, 6000, 10000, 10000]))
            depth = float(rng.choice([10, 20, 30, 30, 40]))            # reads x read length / genome
            glen = 200_000
            fk = dict(n_reads=int(depth * glen / rl), read_len=rl, window_len=int(rng.choice([300, 500, 500, 500, 800])), sub=kw["sub"] if not short else 0.03,
                      ins=kw["ins"] if not short else 0.03, dele=kw["dele"] if not short else 0.04, min_overlap=int(rng.choice([500, 2000])))
            kw = dict(fk, coverage=depth); w = fk["window_len"]
            jobs = [(glen, dict(fk, seed=int(rng.integers(1, 2**31)))) for _ in range(a.workers)]
        tg = time.time()
        with mp.get_context("fork").Pool(min(a.workers, len(jobs))) as pool:
            parts = pool.map(piece, jobs)
        b = parts[0]
        for p in parts[1:]:
            b = b.concat(p)
        t_gen = time.time() - tg
        key = (scores, trim)
        if key not in engines:
            engines[key] = HipEngine(*scores, trim)
        tg = time.time(); g = engines[key].consensus(b); t_gpu = time.time() - tg
        tg = time.time(); o = oracle_lib.consensus(b, *scores, trim, 0, simd=True); t_cpu = time.time() - tg
        bad = [i for i in range(b.n_windows) if g.consensus[i] != o.consensus[i] or int(g.polished[i]) != int(o.polished[i])]
        for i in bad[:4]:
            b.select([i]).save(os.path.join(a.out, "differ_round%d_window%d.npz" % (k, i)))
        total += b.n_windows; differ += len(bad)
        if bad and len(first) < 10:
            first.append({"round": k, "windows": bad[:5], "workload": kw, "scores": scores, "trim": trim})
        rounds.append({"round": k, "windows": b.n_windows, "differ": len(bad), "w": w, "coverage": kw["coverage"], "read_len": kw["read_len"], "sub_ins_del": [kw["sub"], kw["ins"], kw["dele"]],
                       "backbone_errors": kw.get("backbone_errors"), "quality": kw.get("with_quality"), "fragment": bool(a.fragment), "low_complexity": kw.get("low_complexity"), "n_rate": kw.get("n_rate"), "scores": scores, "trim": trim, "s_generate_gpu_cpu": [round(t_gen, 1), round(t_gpu, 1), round(t_cpu, 1)]})
        sys.stderr.write("round %d: %d windows (w %d, %gx, reads %d), %d differ; %.0f s so far\n" % (k, b.n_windows, w, kw["coverage"], kw["read_len"], len(bad), time.time() - t0)); sys.stderr.flush()
    print(json.dumps({"what": "synthetic windows of drawn shapes, HIP engine against the CPU oracle, every window", "rounds": len(rounds), "windows": total, "windows_differ": differ,
                      "first": first, "seconds": round(time.time() - t0, 1), "per_round": rounds}))


if __name__ == "__main__":
    main()
