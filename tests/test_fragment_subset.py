"""tools/cfg5_whole.py checks BASELINE configs[4] at full size against the oracle without a second full run: a sample of TARGETS with every
overlap onto them is polished as a job of its own (host layer + host aligner + oracle) and each FASTA record must equal the record the whole
job printed for that read.  That only proves something if the sub-job reproduces the whole job's records exactly -- in -f mode a read's windows
hold the overlaps onto it and nothing else (reference src/polisher.cpp:295 keeps every overlap per query, :405-461 cuts layers per overlap).
Here, on the CPU, at a size the host aligner finishes in seconds: whole job and sub-job through the same host layer + oracle."""
import os

import numpy as np

from racon_amd.synth import simulate_fragment_files


def _records(fasta: bytes):
    lines = fasta.split(b"\n")
    return {lines[i][1:].split(b" ", 1)[0]: (lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 2)}


def test_a_sample_of_targets_reproduces_their_records_of_the_whole_job(tmp_path, oracle):
    from racon_amd.polisher import Polisher
    d = str(tmp_path)
    n_reads = 120
    p = simulate_fragment_files(d, 40_000, n_reads, seed=20260924)

    def run(reads, paf, targets):
        P = Polisher(reads, paf, targets, "kF", 500, 10.0, 0.3, True, 3, -5, -4, 8, 1)
        P.initialize()
        b = P.windows()
        fa = P.assemble(oracle.consensus(b, 3, -5, -4, True, 0, simd=True), True)
        P.close()
        return _records(fa), b.n_windows

    whole, nw = run(p["reads"], p["paf"], p["reads"])
    assert len(whole) > n_reads // 2 and nw > 1000
    rng = np.random.default_rng(7)
    names = {b"f%d" % i for i in rng.choice(n_reads, 12, replace=False).tolist()}
    sub_t, sub_p = os.path.join(d, "t.fastq"), os.path.join(d, "s.paf")
    with open(p["reads"], "rb") as f, open(sub_t, "wb") as ft:
        while True:
            h = f.readline()
            if not h:
                break
            rec = [h, f.readline(), f.readline(), f.readline()]
            if h[1:].rstrip(b"\n") in names:
                ft.writelines(rec)
    with open(p["paf"], "rb") as f, open(sub_p, "wb") as fp:
        for line in f:
            if line.split(b"\t", 6)[5] in names:
                fp.write(line)
    sub, nws = run(p["reads"], sub_p, sub_t)
    assert 0 < len(sub) <= len(names) and nws < nw
    for name, rec in sub.items():
        assert whole.get(name) == rec, name            # header (LN / RC / XC tags) and sequence
