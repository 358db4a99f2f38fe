#!/usr/bin/env python
"""Generates tests/golden/* from the reference's own test data (run in the build
container, where /root/reference exists; the GPU box only sees the committed output).

For every parameterisation of the reference's golden tests that isolates the window-consensus
path (reference test/racon_test.cpp:133-177: SAM overlaps, no pre-alignment) plus one
PAF / W=1000 / edit-distance-score case and a slice of the fragment-correction set, it stores:
  * the packed windows the host layer (racon_amd/host, Polisher::initialize) builds  -> <name>.npz
  * the oracle's per-window consensus + flags for those windows, the polished-contig md5 and
    the golden number of the reference test (edit distance to sample_reference) -> manifest.json
  * the reference contig (sample_reference.fasta.gz, data not source)           -> reference_contig.txt.gz
The oracle results stored here are only trusted because the same run asserts the reference's
golden numbers (racon_test.cpp:104,128,151,174,197,220) on them.
"""
import gzip
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from racon_amd import polisher as P  # noqa: E402
from oracle import oracle_lib  # noqa: E402

D = "/root/reference/test/data/"
OUT = os.path.join(ROOT, "tests", "golden")


def revcomp(s: bytes) -> bytes:
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = b"".join(gzip.open(D + "sample_reference.fasta.gz").read().split(b"\n")[1:])
    with gzip.open(os.path.join(OUT, "reference_contig.txt.gz"), "wb", compresslevel=9) as f:
        f.write(ref)
    cases = [  # name, reads, overlaps, target, type, w, scores, golden ED (racon_test.cpp line), keep windows
        ("sam_fastq_w500", "sample_reads.fastq.gz", "sample_overlaps.sam.gz", "sample_layout.fasta.gz", "kC", 500, (5, -4, -8), 1317, None),
        ("sam_fasta_w500", "sample_reads.fasta.gz", "sample_overlaps.sam.gz", "sample_layout.fasta.gz", "kC", 500, (5, -4, -8), 1770, None),
        ("paf_fastq_w1000", "sample_reads.fastq.gz", "sample_overlaps.paf.gz", "sample_layout.fasta.gz", "kC", 1000, (5, -4, -8), 1289, None),
        ("paf_fastq_w500_edit", "sample_reads.fastq.gz", "sample_overlaps.paf.gz", "sample_layout.fasta.gz", "kC", 500, (1, -1, -1), 1321, None),
        ("frag_kF_fastq_first200", "sample_reads.fastq.gz", "sample_ava_overlaps.paf.gz", "sample_reads.fastq.gz", "kF", 500, (1, -1, -1), None, 200),
    ]
    manifest = {"generator": "tools/make_golden.py", "source": "reference test/data (racon v1.5.0)", "cases": {}}
    for name, reads, ovl, tgt, ty, w, sc, gold, keep in cases:
        p = P.Polisher(D + reads, D + ovl, D + tgt, ty, w, 10, 0.3, True, *sc, num_threads=8)
        p.initialize()
        b = p.windows()
        res = oracle_lib.consensus(b, *sc, True, 0)
        entry = {"scores": list(sc), "window_length": w, "type": ty, "trim": True, "n_windows_total": b.n_windows}
        if gold is not None:
            fa = P.parse_fasta(p.assemble(res, True))
            assert len(fa) == 1
            ed = P.edit_distance(revcomp(fa[0][1]), ref)
            assert ed == gold, (name, ed, gold)
            # contig polishing of ONE target: the polished sequence is the concatenation of the window consensi
            assert b"".join(res.consensus) == fa[0][1]
            entry.update({"golden_edit_distance": gold, "polished_md5": hashlib.md5(fa[0][1]).hexdigest(),
                          "header": fa[0][0].decode()})
        if keep is not None:
            b = b.select(range(keep))
            res = oracle_lib.consensus(b, *sc, True, 0)
        b.save(os.path.join(OUT, name + ".npz"))
        entry.update({"n_windows": b.n_windows,
                      "consensus_md5": [hashlib.md5(c).hexdigest() for c in res.consensus],
                      "consensus_len": [len(c) for c in res.consensus],
                      "polished": [int(v) for v in res.polished], "chimeric": [int(v) for v in res.chimeric]})
        manifest["cases"][name] = entry
        print(name, b.n_windows, "windows", os.path.getsize(os.path.join(OUT, name + ".npz")) >> 10, "KiB", flush=True)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
