#!/usr/bin/env python
"""cfg5 WHOLE, every record (no GPU needed): the FASTA the drop-in binary printed for BASELINE.json configs[4] at scale 1 (tools/cfg5_whole.py
--record-md5: one md5 per read's record) against host layer (host aligner = the edlib-equivalent) + CPU oracle on ALL 100 000 targets.
The input files are regenerated from the same seed (racon_amd.synth.simulate_fragment_files is deterministic); the targets are taken in
chunks, each with every overlap onto it as a job of its own (in -f mode a read's windows only hold the overlaps onto it:
tests/test_fragment_subset.py holds that premise), so memory stays bounded and the check can run beside other work.  Resumable: finished
chunks are kept in --state.  Prints one JSON line at the end (and a progress line per chunk on stderr)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                      # noqa: E402
from racon_amd.synth import simulate_fragment_files                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--md5", required=True, help=".npy written by tools/cfg5_whole.py --record-md5 on the GPU box")
ap.add_argument("--chunk", type=int, default=1000, help="targets per chunk")
ap.add_argument("--threads", type=int, default=7)
ap.add_argument("--dir", default=os.environ.get("RACON_AMD_CACHE", "/tmp/racon_amd_cache"))
ap.add_argument("--state", default="")
ap.add_argument("--max-chunks", type=int, default=0)
a = ap.parse_args()

d = os.path.join(a.dir, "cfg5_%g" % a.scale)
t0 = time.time()
done = os.path.join(d, ".done")
if not os.path.exists(done):
    p = simulate_fragment_files(d, int(33_333_333 * a.scale), int(100_000 * a.scale), seed=20260924)
    open(done, "w").write(str(p["n_overlaps"]))
reads, paf = os.path.join(d, "reads.fastq"), os.path.join(d, "overlaps.paf")
n_targets = int(100_000 * a.scale)
want = np.load(a.md5)
assert want.shape == (n_targets, 16), want.shape
sys.stderr.write("files ready after %.0f s\n" % (time.time() - t0))

# ---- the targets and the overlaps onto them, chunk by chunk, as files of their own
n_chunks = (n_targets + a.chunk - 1) // a.chunk
cdir = os.path.join(d, "full_check_%d" % a.chunk)
if not os.path.exists(os.path.join(cdir, ".split")):
    os.makedirs(cdir, exist_ok=True)
    ft = [open(os.path.join(cdir, "t%04d.fastq" % c), "wb") for c in range(n_chunks)]
    with open(reads, "rb") as f:
        i = 0
        while True:
            h = f.readline()
            if not h:
                break
            rec = [h, f.readline(), f.readline(), f.readline()]
            assert h[1:].rstrip(b"\n") == b"f%d" % i
            ft[i // a.chunk].writelines(rec)
            i += 1
    for x in ft:
        x.close()
    fp = [open(os.path.join(cdir, "o%04d.paf" % c), "wb") for c in range(n_chunks)]
    with open(paf, "rb") as f:
        for line in f:
            fp[int(line.split(b"\t", 6)[5][1:]) // a.chunk].write(line)
    for x in fp:
        x.close()
    open(os.path.join(cdir, ".split"), "w").write("ok")
    sys.stderr.write("split into %d chunks after %.0f s\n" % (n_chunks, time.time() - t0))

from racon_amd.polisher import Polisher                                  # noqa: E402
from oracle import oracle_lib                                            # noqa: E402

state_path = a.state or os.path.join(cdir, "state.json")
state = json.load(open(state_path)) if os.path.exists(state_path) else {"chunks": {}}
t_run = time.time()
n_new = 0
for c in range(n_chunks):
    if str(c) in state["chunks"]:
        continue
    if a.max_chunks and n_new >= a.max_chunks:
        break
    tc = time.time()
    sub_t, sub_p = os.path.join(cdir, "t%04d.fastq" % c), os.path.join(cdir, "o%04d.paf" % c)
    p = Polisher(reads, sub_p, sub_t, "kF", 500, 10.0, 0.3, True, 3, -5, -4, a.threads, 1)
    p.initialize()
    b = p.windows()
    ref = oracle_lib.consensus(b, 3, -5, -4, True, 0, simd=True)
    fasta = p.assemble(ref, True)          # (the binary drops reads without a polished window unless -u: same here)
    p.close()
    lo = c * a.chunk
    got = np.zeros((min(a.chunk, n_targets - lo), 16), np.uint8)
    lines = fasta.split(b"\n")
    for k in range(0, len(lines) - 1, 2):
        h, sq = lines[k][1:], lines[k + 1]
        got[int(h.split(b" ", 1)[0][1:].rstrip(b"r")) - lo] = np.frombuffer(hashlib.md5(h + b"\n" + sq).digest(), np.uint8)
    w = want[lo:lo + got.shape[0]]
    bad = np.nonzero((got != w).any(axis=1))[0]
    state["chunks"][str(c)] = {"targets": int(got.shape[0]), "records": int((got.any(axis=1)).sum()), "records_in_the_gpu_fasta": int((w.any(axis=1)).sum()),
                               "windows": int(b.n_windows), "differ": int(bad.size), "first": [int(lo + x) for x in bad[:5]], "seconds": round(time.time() - tc, 1)}
    json.dump(state, open(state_path, "w"))
    n_new += 1
    sys.stderr.write("chunk %d / %d: %d windows, %d records, %d differ, %.0f s (%.0f s so far)\n" % (c + 1, n_chunks, b.n_windows, state["chunks"][str(c)]["records"], bad.size,
                                                                                                 time.time() - tc, time.time() - t_run))
    sys.stderr.flush()
ch = state["chunks"]
out = {"workload": "cfg5 (fragment correction, -f) at scale %g: every FASTA record of the one-GPU run against host layer (host aligner) + CPU oracle" % a.scale,
       "chunks_done": len(ch), "chunks": n_chunks, "targets": sum(v["targets"] for v in ch.values()), "records": sum(v["records"] for v in ch.values()),
       "records_in_the_gpu_fasta": sum(v["records_in_the_gpu_fasta"] for v in ch.values()), "windows": sum(v["windows"] for v in ch.values()),
       "records_differ": sum(v["differ"] for v in ch.values()), "first": sum((v["first"] for v in ch.values()), [])[:10],
       "cpu_seconds_wall": round(sum(v["seconds"] for v in ch.values()), 0), "threads": a.threads, "md5_file": os.path.basename(a.md5),
       "ok": len(ch) == n_chunks and all(v["differ"] == 0 for v in ch.values())}
print(json.dumps(out))
