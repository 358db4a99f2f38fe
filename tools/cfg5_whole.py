#!/usr/bin/env python
"""BASELINE.json configs[4] WHOLE on one GPU: fragment correction, `racon -f reads overlaps reads`, 100 000 x 10 kbp reads with dual
overlaps (~4.8 M overlaps that all need the pre-alignment of reference src/overlap.cpp:205-224, ~2 M windows; reference
src/polisher.cpp:295 keeps every overlap per query in kF mode, :388-461 cuts the windows) through the drop-in binary with everything
on the device (`--cudaaligner-batches 1`: alignment, breaking points, windows, consensus) in --shards sequential window ranges
(RACON_HIP_DEVICE_SHARDS: what eight GPUs would each take, one after the other on the one device).

  1. the input files (racon_amd.synth.simulate_fragment_files);
  2. the binary: wall clock, the Logger's stage times, per-shard timing lines, peak HBM in use, FASTA md5;
  3. parity at size without a second full run: a seeded sample of TARGETS (--sample of the reads) with every overlap onto them, as files of
     their own -> host layer (host aligner = the edlib-equivalent) + CPU oracle -> FASTA records; each must equal the record the big run
     printed for that read (in -f mode a read's windows only hold the overlaps onto it, so the sub-job reproduces them exactly);
  4. the sub-job through the binary with the device aligner and with device-built windows from host CIGARs: same FASTA as 3.
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
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--sample", type=float, default=0.01, help="fraction of the reads whose records are checked against host layer + oracle")
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--dir", default=os.environ.get("RACON_AMD_CACHE", "/tmp/racon_amd_cache"))
ap.add_argument("--keep-fasta", default="")
ap.add_argument("--record-md5", default="", help="write the md5 of every FASTA record of the big run (one row per read, zeros = no record) as .npy: tools/cfg5_full_check.py compares ALL of them with host layer + oracle on a CPU box")
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
            "files_s": round(time.time() - t, 1), "dir": d}


def cli(reads, paf, targets, flags=(), env_add=None, keep=None):
    env = dict(os.environ)
    for k in ("RACON_HIP_DEVICE_WINDOWS", "RACON_HIP_DEVICE_SHARDS"):
        env.pop(k, None)
    env.update(env_add or {})
    env["RACON_HIP_TIMING"] = "1"
    t = time.time()
    out_path = keep or os.path.join(a.dir, "cfg5_whole_out_%d.fasta" % os.getpid())
    with open(out_path, "wb") as fo:
        r = subprocess.run([EXE, "-f", "-t", str(a.threads)] + list(flags) + [reads, paf, targets], stdout=fo, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t
    err = r.stderr.decode(errors="replace")
    stages = {}
    for m in re.finditer(r"\[racon::Polisher::(\w*)\] ([^\n\r]*?) (\d+\.\d+) s", err):
        stages[(m.group(1) + " " + m.group(2)).strip()] = float(m.group(3))
    h = hashlib.md5()
    with open(out_path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    shard_lines = [l for l in err.splitlines() if "timing: shard" in l or "peak HBM" in l]
    peak = re.search(r"peak HBM in use (\d+\.\d+) GB", err)
    return {"rc": r.returncode, "wall_s": round(wall, 2), "stages_s": stages, "fasta_bytes": os.path.getsize(out_path), "md5": h.hexdigest(),
            "shard_timing": [l.split("timing: ", 1)[1][:220] for l in shard_lines][:12], "peak_hbm_gb": float(peak.group(1)) if peak else None,
            "stderr_tail": err[-400:] if r.returncode else ""}, out_path


def records(path):
    """FASTA file -> {read name: (header, sequence)} (one line per sequence, as racon prints it)."""
    out = {}
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\n")
            h = h.rstrip(b"\n")[1:]
            name = h.split(b" ", 1)[0]
            out[name] = (h, s)
    return out


out = {"workload": "cfg5 (fragment correction, -f) at scale %g: %d reads x 10 kbp, dual overlaps, -w 500, %d sequential window-range shards on one GPU"
                   % (a.scale, int(100_000 * a.scale), a.shards)}
big = files(a.scale)
out["files"] = {"n_overlaps": big["n_overlaps"], "files_s": big["files_s"], "reads_fastq_bytes": os.path.getsize(big["reads"]), "paf_bytes": os.path.getsize(big["paf"])}
run, fasta_path = cli(big["reads"], big["paf"], big["reads"], flags=("--cudaaligner-batches", "1"), env_add={"RACON_HIP_DEVICE_SHARDS": str(a.shards)}, keep=a.keep_fasta or None)
out["device_everything"] = run
if run["rc"] != 0:
    print(json.dumps(out)); sys.exit(1)
big_rec = records(fasta_path)
n_targets = int(100_000 * a.scale)
if a.record_md5:
    dig = np.zeros((n_targets, 16), np.uint8)
    for name, (h, sq) in big_rec.items():
        dig[int(name[1:].rstrip(b"r"))] = np.frombuffer(hashlib.md5(h + b"\n" + sq).digest(), np.uint8)
    os.makedirs(os.path.dirname(os.path.abspath(a.record_md5)), exist_ok=True)
    np.save(a.record_md5, dig)
    out["record_md5_file"] = os.path.basename(a.record_md5)
out["fasta_records"] = len(big_rec)
out["fasta_bases"] = int(sum(len(s) for _, s in big_rec.values()))
# windows of the job: ceil(read length / 500) per read (the targets are the reads)
lens = []
with open(big["reads"], "rb") as f:
    for i, line in enumerate(f):
        if i % 4 == 1:
            lens.append(len(line) - 1)
nw = int(sum((l + 499) // 500 for l in lens))
out["windows"] = nw
pol = run["stages_s"].get("polish generated consensus")
if pol:
    out["windows_per_s_polish_interval"] = nw / pol          # (shards > devices: polish() holds alignment + construction + consensus of every shard)
    out["windows_per_s_whole_binary"] = nw / run["wall_s"]

# ---- parity at size: a sample of targets as a job of its own -> host layer + oracle
rng = np.random.default_rng(20260924)
pick = sorted(rng.choice(n_targets, max(8, int(n_targets * a.sample)), replace=False).tolist())
names = {b"f%d" % i for i in pick}
sub_dir = os.path.join(big["dir"], "subset_%g" % a.sample)
os.makedirs(sub_dir, exist_ok=True)
sub_t, sub_p = os.path.join(sub_dir, "targets.fastq"), os.path.join(sub_dir, "overlaps.paf")
t0 = time.time()
with open(big["reads"], "rb") as f, open(sub_t, "wb") as ft:
    while True:
        h = f.readline()
        if not h:
            break
        rec = [h, f.readline(), f.readline(), f.readline()]
        if h[1:].rstrip(b"\n") in names:
            ft.writelines(rec)
n_sub_ovl = 0
with open(big["paf"], "rb") as f, open(sub_p, "wb") as fp:
    for line in f:
        if line.split(b"\t", 6)[5] in names:
            fp.write(line); n_sub_ovl += 1
out["subset"] = {"targets": len(pick), "overlaps": n_sub_ovl, "files_s": round(time.time() - t0, 1)}

from racon_amd.polisher import Polisher                                  # noqa: E402
from oracle import oracle_lib                                            # noqa: E402
t0 = time.time()
p = Polisher(big["reads"], sub_p, sub_t, "kF", 500, 10.0, 0.3, True, 3, -5, -4, a.threads, 1)
p.initialize()
b = p.windows()
ref = oracle_lib.consensus(b, 3, -5, -4, True, 0, simd=True)
ref_fasta = p.assemble(ref, True)           # (the binary drops reads without a polished window unless -u: same here)
p.close()
ref_path = os.path.join(sub_dir, "oracle.fasta")
open(ref_path, "wb").write(ref_fasta)
ref_rec = records(ref_path)
bad = [n.decode() for n, v in ref_rec.items() if big_rec.get(n) != v]
out["oracle_sample"] = {"targets": len(ref_rec), "windows": int(b.n_windows), "records_differ": len(bad), "first": bad[:5],
                        "host_layer_and_oracle_s": round(time.time() - t0, 1),
                        "what": "host layer (host aligner) + CPU oracle on the sampled targets with every overlap onto them; each FASTA record (header with LN/RC/XC tags + sequence) "
                                "must equal the record the whole job printed for that read"}
# ... and the sub-job through the binary: device aligner (the product flag) and device-built windows from host CIGARs
sub_runs = {}
for label, flags, env_add in (("device_align", ("--cudaaligner-batches", "1"), {}), ("device_cigars_host_aligner", (), {"RACON_HIP_DEVICE_WINDOWS": "2"})):
    r, pth = cli(big["reads"], sub_p, sub_t, flags=flags, env_add=env_add, keep=os.path.join(sub_dir, label + ".fasta"))
    # (-f prints unpolished reads too only with -u; the oracle FASTA above was assembled with drop_unpolished = False: compare records)
    rr = records(pth)
    r["records_equal_oracle"] = all(ref_rec.get(n) == v for n, v in rr.items()) and len(rr) > 0
    r["records"] = len(rr)
    sub_runs[label] = {k: r[k] for k in ("rc", "wall_s", "md5", "records", "records_equal_oracle")}
out["subset_through_the_binary"] = sub_runs
out["ok"] = bool(run["rc"] == 0 and not bad and all(v["records_equal_oracle"] for v in sub_runs.values()))
if not a.keep_fasta:
    try:
        os.remove(fasta_path)
    except OSError:
        pass
print(json.dumps(out))
