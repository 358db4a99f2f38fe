#!/usr/bin/env python
"""Wall clock of the whole drop-in binary on cfg2-shaped FILES (1 Mbp contig, 30x of 10 kb ONT-like reads, gz FASTQ +
SAM / PAF + gz FASTA): `racon_hip` with the stages on the host or on the device, the Logger's own stage times
(reference src/logger.cpp:20-54 prints the same lines), and a byte comparison of the FASTA of every mode.
  mode 0: host parses, aligns (PAF), walks CIGARs, cuts windows; device polishes          (the round-1 product)
  mode 2: windows and the CIGAR walk on the device (SAM: nothing but parsing left on the host)
  mode 3: PAF: the pairwise alignment on the device as well (rcn_engine_build_windows_from_pairs)
Prints one JSON line; tools/gpu_round.sh keeps it under profiles/."""
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
from racon_amd.synth import simulate_files  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--contig", type=int, default=1_000_000)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--dir", default="/tmp/e2e_cfg2")
a = ap.parse_args()

t = time.time()
paths, _ = simulate_files(a.dir, contig_len=a.contig, coverage=30.0, read_len=10000, n_contigs=1, seed=5)
out = {"workload": f"cfg2-shaped files: {a.contig} bp contig, 30x, 10 kb reads; racon_hip -t {a.threads}", "simulate_s": round(time.time() - t, 1), "runs": []}
exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
digests = {}
for ovl, mode, serial in [("sam", "0", "1"), ("sam", "0", ""), ("sam", "2", ""), ("paf", "0", ""), ("paf", "3", "")]:
    env = dict(os.environ)
    env.pop("RACON_HIP_DEVICE_WINDOWS", None); env.pop("RACON_HIP_SERIAL_INGEST", None)
    if mode != "0":
        env["RACON_HIP_DEVICE_WINDOWS"] = mode
    if serial:
        env["RACON_HIP_SERIAL_INGEST"] = "1"
    t = time.time()
    r = subprocess.run([exe, "-t", str(a.threads), paths["reads"], paths[ovl], paths["targets"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t
    stages = {}
    for m in re.finditer(r"\[racon::Polisher::(\w*)\] ([^\n\r]*?) (\d+\.\d+) s", r.stderr.decode(errors="replace")):
        stages[(m.group(1) + " " + m.group(2)).strip()] = float(m.group(3))
    digests.setdefault(ovl, set()).add(hashlib.md5(r.stdout).hexdigest())
    out["runs"].append({"overlaps": ovl, "device_windows_mode": int(mode), "serial_ingest": bool(serial), "rc": r.returncode,
                        "wall_s": round(wall, 2), "fasta_bytes": len(r.stdout), "stages_s": stages})
out["fasta_identical_across_modes"] = {k: len(v) == 1 for k, v in digests.items()}
print(json.dumps(out))
