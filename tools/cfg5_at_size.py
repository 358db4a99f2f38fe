#!/usr/bin/env python
"""BASELINE.json configs[4] -- fragment correction, `racon -f reads overlaps reads`, 100 000 x 10 kbp reads with dual
overlaps, 2 M windows on 8 GPUs -- at ONE GPU's share of it (--scale 0.125: 12 500 reads, ~250 000 windows, ~600 000
overlaps that all need the pre-alignment of reference src/overlap.cpp:205-224) through the drop-in binary:

  1. the input files (racon_amd.synth.simulate_fragment_files: FASTQ + PAF with both directions of every pair);
  2. `racon_hip -f` with everything on the device (RACON_HIP_DEVICE_WINDOWS=3: alignment, breaking points, windows,
     consensus): wall clock, the Logger's stage times, windows/s over the polish() interval;
  3. parity at size: the windows the device built are copied back (rcn_engine_export_batch) and a seeded sample of them
     (--sample, default 5 %) is compared with the CPU oracle, window by window; the consensus of EVERY window must equal
     the FASTA the binary printed;
  4. at --cross-scale (default 0.01) additionally the binary with the host doing the alignment (mode 0, the edlib-equivalent
     on the host's cores) and with device-built windows from host CIGARs (mode 2): the three FASTA files must be identical
     (the host aligner at full share would take minutes of the box's 16 cores; its CIGARs equal the device's on every
     overlap of the reference's own PAF samples, tests/test_gpu_pair_align.py).
Prints one JSON line."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                      # noqa: E402
from racon_amd.synth import simulate_fragment_files                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.125)
ap.add_argument("--cross-scale", type=float, default=0.01)
ap.add_argument("--host-at-size", action="store_true", help="also the binary with the HOST aligner on the full share (minutes of CPU): the two FASTA files must be identical")
ap.add_argument("--sample", type=float, default=0.05)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--dir", default=os.environ.get("RACON_AMD_CACHE", "/tmp/racon_amd_cache"))
a = ap.parse_args()
EXE = os.path.join(ROOT, "racon_amd", "host", "racon_hip")


def files(scale):
    d = os.path.join(a.dir, "cfg5_%g" % scale)
    t = time.time()
    done = os.path.join(d, ".done")
    if not os.path.exists(done):
        p = simulate_fragment_files(d, int(33_333_333 * scale), int(100_000 * scale), seed=20260924)
        open(done, "w").write(str(p["n_overlaps"]))
    return {"reads": os.path.join(d, "reads.fastq"), "paf": os.path.join(d, "overlaps.paf"), "n_overlaps": int(open(done).read()),
            "files_s": round(time.time() - t, 1)}


def cli(paths, mode):
    env = dict(os.environ)
    env.pop("RACON_HIP_DEVICE_WINDOWS", None)
    if mode != "0":
        env["RACON_HIP_DEVICE_WINDOWS"] = mode
    t = time.time()
    r = subprocess.run([EXE, "-f", "-t", str(a.threads), paths["reads"], paths["paf"], paths["reads"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t
    stages = {}
    for m in re.finditer(r"\[racon::Polisher::(\w*)\] ([^\n\r]*?) (\d+\.\d+) s", r.stderr.decode(errors="replace")):
        stages[(m.group(1) + " " + m.group(2)).strip()] = float(m.group(3))
    return {"mode": int(mode), "rc": r.returncode, "wall_s": round(wall, 2), "stages_s": stages, "fasta_bytes": len(r.stdout),
            "md5": hashlib.md5(r.stdout).hexdigest(), "stderr_tail": r.stderr.decode(errors="replace")[-300:] if r.returncode else ""}, r.stdout


out = {"workload": "cfg5 (fragment correction, -f) at scale %g: %d reads x 10 kbp, dual overlaps, -w 500" % (a.scale, int(100_000 * a.scale))}
big = files(a.scale)
out["files"] = {"n_overlaps": big["n_overlaps"], "files_s": big["files_s"]}
run3, fasta3 = cli(big, "3")
out["device_everything"] = run3

# ---- parity at size: the device-built windows, a sample against the oracle, all of them against the FASTA
from racon_amd.engine import HipEngine                                  # noqa: E402
from racon_amd.polisher import Polisher                                  # noqa: E402
from oracle import oracle_lib                                            # noqa: E402
os.environ["RACON_HIP_DEVICE_WINDOWS"] = "3"
p = Polisher(big["reads"], big["paf"], big["reads"], "kF", 500, 10.0, 0.3, True, 3, -5, -4, a.threads, 1)
p.initialize(keep_layout=True)
del os.environ["RACON_HIP_DEVICE_WINDOWS"]
reads, _, wt, wl, qt = p.layout()
pairs = p.pairs()
eng = HipEngine(3, -5, -4, True)
t = time.time()
eng.build_windows_from_pairs(reads, pairs, wl, qt, wt)
res = eng.run()
t_dev = time.time() - t
b = eng.export_batch()
nw = b.n_windows
out["windows"] = nw
out["layers"] = int(b.n_seqs - nw)
out["engine_build_and_polish_s"] = round(t_dev, 2)
pol = run3["stages_s"].get("polish generated consensus")
if pol:
    out["windows_per_s_polish_interval"] = nw / pol          # (in this mode polish() holds alignment + construction + consensus)
st = eng.stats()
out["consensus_kernel"] = {"kernel_ms": st["kernel_ms"], "windows_per_s": nw / (st["kernel_ms"] / 1e3), "dp_bytes": st["dp_bytes"],
                           "roofline_frac_of_8TBs": st["dp_bytes"] / (st["kernel_ms"] / 1e3) / 8e12, "n_retried": st["n_retried"]}
# every window's consensus is what the binary printed (targets in order, windows in order; a read whose windows were all
# left unpolished is dropped from the output, reference src/polisher.cpp:513-520 -- compare per target)
seqs = fasta3.split(b"\n")[1::2]
tl = np.diff(reads.seq_off.astype(np.int64))[:int(reads.n_targets)]
wpt = (tl + wl - 1) // wl
first = np.concatenate([[0], np.cumsum(wpt)])
k = 0; same = True; dropped = 0
for tgt in range(int(reads.n_targets)):
    w0, w1 = int(first[tgt]), int(first[tgt + 1])
    if not res.polished[w0:w1].any():
        dropped += 1
        continue
    if k >= len(seqs) or seqs[k] != b"".join(res.consensus[w0:w1]):
        same = False
        break
    k += 1
out["fasta_equals_engine_consensus"] = bool(same and k == len([s for s in seqs if s]))
out["targets_dropped"] = dropped
rng = np.random.default_rng(20260924)
pick = sorted(rng.choice(nw, max(64, int(nw * a.sample)), replace=False).tolist())
sub = b.select(pick)
t = time.time()
ref = oracle_lib.consensus(sub, 3, -5, -4, True, 0, simd=True)
bad = [pick[i] for i in range(len(pick)) if ref.consensus[i] != res.consensus[pick[i]] or ref.polished[i] != res.polished[pick[i]]]
out["oracle_sample"] = {"windows": len(pick), "differ": len(bad), "first": bad[:5], "oracle_s": round(time.time() - t, 1)}
p.close()

# ---- the three ways through the binary on a small share: same FASTA
if a.host_at_size:
    run0, _ = cli(big, "0")
    out["host_aligner_at_size"] = {"run": run0, "identical_to_device_everything": run0["md5"] == run3["md5"]}
if a.cross_scale > 0:
    small = files(a.cross_scale)
    runs = [cli(small, m)[0] for m in ("0", "2", "3")]
    out["cross_check"] = {"scale": a.cross_scale, "n_overlaps": small["n_overlaps"], "runs": runs, "identical": len({r["md5"] for r in runs}) == 1}
print(json.dumps(out))
