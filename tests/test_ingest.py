"""Parallel ingest (racon_amd/host/parsers.cpp read_batches; SURVEY 8(f) rank 3, reference src/polisher.cpp:200-349):
one inflating thread per file frames records, worker threads build Sequence / Overlap objects, the three input files are
read concurrently.  What comes out of Polisher::initialize must not depend on any of that: the windows are compared, array
for array, with a serial run (RACON_HIP_SERIAL_INGEST=1) on the reference's data and on hand-made files that hit the
framing corners (multi-line FASTQ whose quality lines start with '@' or '+', CRLF, blank lines, a missing final newline,
records straddling the 4 MiB inflate blocks, plain and gzip)."""
import gzip
import os
import time

import numpy as np
import pytest

from helpers import REFDATA as DATA


@pytest.fixture(scope="module")
def P():
    from racon_amd import polisher
    polisher.build()
    return polisher


def _windows(P, reads, ovl, targets, typ, threads, serial, monkeypatch):
    if serial:
        monkeypatch.setenv("RACON_HIP_SERIAL_INGEST", "1")
    else:
        monkeypatch.delenv("RACON_HIP_SERIAL_INGEST", raising=False)
    p = P.Polisher(reads, ovl, targets, typ, 500, 10, 0.3, True, 3, -5, -4, num_threads=threads)
    t = time.perf_counter()
    p.initialize()
    dt = time.perf_counter() - t
    return p.windows(), dt


def _same(a, b):
    for f in ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("reads,ovl,targets,typ", [
    ("sample_reads.fastq.gz", "sample_overlaps.sam.gz", "sample_layout.fasta.gz", "kC"),
    ("sample_reads.fasta.gz", "sample_overlaps.paf.gz", "sample_layout.fasta.gz", "kC"),
    ("sample_reads.fastq.gz", "sample_ava_overlaps.mhap.gz", "sample_reads.fastq.gz", "kF"),
])
def test_parallel_ingest_gives_the_serial_windows(P, reads, ovl, targets, typ, monkeypatch):
    ws, ts = _windows(P, DATA + reads, DATA + ovl, DATA + targets, typ, 8, True, monkeypatch)
    wp, tp = _windows(P, DATA + reads, DATA + ovl, DATA + targets, typ, 8, False, monkeypatch)
    _same(ws, wp)
    print("initialize(): serial ingest %.2f s, parallel %.2f s" % (ts, tp))


def test_framing_corner_cases(P, tmp_path, monkeypatch):
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    contig = acgt[rng.integers(0, 4, 60000)].tobytes()
    # reads: exact substrings (so that the PAF pre-alignment is trivial), written as multi-line FASTQ with nasty qualities
    reads, recs = [], []
    for k in range(900):
        a = int(rng.integers(0, 60000 - 6000)); n = int(rng.integers(3000, 6000))
        s = contig[a:a + n]
        q = bytearray((rng.integers(0, 30, n) + 33).astype(np.uint8).tobytes())
        q[0:1] = b"@"                                   # a quality line that starts with '@'
        width = int(rng.integers(50, 120))
        q[width:width + 1] = b"+"                       # ... and one that starts with '+'
        q[2 * width:2 * width + 1] = b"@"
        reads.append((b"read%d extra words" % k, s, bytes(q), a, n, width))
    fq = bytearray()
    for name, s, q, a, n, width in reads:
        eol = b"\r\n" if len(fq) % 3 == 0 else b"\n"
        fq += b"@" + name + eol
        for i in range(0, n, width):
            fq += s[i:i + width] + eol
        fq += b"+" + eol
        for i in range(0, n, width):
            fq += q[i:i + width] + eol
        if len(fq) % 5 == 0:
            fq += b"\n"                                 # blank line between records
    assert len(fq) > 6 << 20                            # several 4 MiB inflate blocks
    fq = bytes(fq).rstrip(b"\r\n")                      # no final newline
    paf = b"".join(b"read%d\t%d\t0\t%d\t+\tctg\t60000\t%d\t%d\t%d\t%d\t60\n" % (k, n, n, a, a + n, n, n)
                   for k, (_, _, _, a, n, _) in enumerate(reads))
    fa = b">ctg some description\n" + b"\n".join(contig[i:i + 70] for i in range(0, len(contig), 70))   # multi-line, no final newline
    paths = {}
    for ext, blob in (("reads.fastq", fq), ("ovl.paf", paf), ("ctg.fasta", fa)):
        plain = tmp_path / ext
        plain.write_bytes(blob)
        with gzip.open(str(plain) + ".gz", "wb") as f:
            f.write(blob)
        paths[ext] = str(plain)
    ws, _ = _windows(P, paths["reads.fastq"], paths["ovl.paf"], paths["ctg.fasta"], "kC", 6, True, monkeypatch)
    wp, _ = _windows(P, paths["reads.fastq"], paths["ovl.paf"], paths["ctg.fasta"], "kC", 6, False, monkeypatch)
    wz, _ = _windows(P, paths["reads.fastq"] + ".gz", paths["ovl.paf"] + ".gz", paths["ctg.fasta"] + ".gz", "kC", 6, False, monkeypatch)
    _same(ws, wp)
    _same(ws, wz)
    # uncompressed files are memory-mapped and framed in place (round 6); through zlib's transparent read instead: the same windows
    monkeypatch.setenv("RACON_HIP_NO_MMAP", "1")
    wn, _ = _windows(P, paths["reads.fastq"], paths["ovl.paf"], paths["ctg.fasta"], "kC", 6, False, monkeypatch)
    monkeypatch.delenv("RACON_HIP_NO_MMAP")
    _same(ws, wn)
    # and the content is what was written: every layer is an exact stretch of the contig with its own qualities
    assert ws.n_windows == 120
    want = {s: q for _, s, q, _, _, _ in reads}
    n_layers = int(ws.win_seq_off[-1]) - ws.n_windows
    assert n_layers > 5000
    w0 = ws.window(0)
    for sq in w0["seqs"][1:4]:
        assert sq[0] in contig


def test_malformed_inputs_are_reported(P, tmp_path):
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r1\nACGT\n+\nIII\n")             # quality shorter than the bases
    good = tmp_path / "t.fasta"
    good.write_bytes(b">t\nACGTACGT\n")
    ovl = tmp_path / "o.paf"
    ovl.write_bytes(b"r1\t4\t0\t4\t+\tt\t8\t0\t4\t4\t4\t60\n")
    p = P.Polisher(str(bad), str(ovl), str(good), "kC", 500, 10, 0.3, True, 3, -5, -4, num_threads=4)
    with pytest.raises(P.RaconError) as e:
        p.initialize()
    assert "invalid FASTQ record" in str(e.value)


def test_records_much_longer_than_a_block(P, tmp_path, monkeypatch):
    """A FASTA record of 40 MB (ten inflate blocks; a chromosome-scale contig is 250 MB) next to small ones: framed once
    per doubling of the carried text, not once per block (parsers.cpp read_batches), with and without line breaks inside
    the record -- the same windows as a serial run, and in time that is linear in the file."""
    rng = np.random.default_rng(9)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    big = acgt[rng.integers(0, 4, 40_000_000)].tobytes()
    small = acgt[rng.integers(0, 4, 3000)].tobytes()
    targets = tmp_path / "targets.fasta"
    with open(targets, "wb") as f:
        f.write(b">small1\n" + small + b"\n>big unwrapped\n" + big + b"\n>big_wrapped\n")
        f.write(b"\n".join(big[k:k + 60_000] for k in range(0, len(big), 60_000)) + b"\n>small2\n" + small[::-1] + b"\n")
    reads = tmp_path / "reads.fasta"
    with open(reads, "wb") as f:
        for k in range(40):
            a = 1000 * k
            f.write(b">r%d\n" % k + big[a:a + 2500] + b"\n")
    paf = tmp_path / "ovl.paf"
    with open(paf, "wb") as f:
        for k in range(40):
            f.write(b"r%d\t2500\t0\t2500\t+\tbig\t%d\t%d\t%d\t2500\t2500\t60\n" % (k, len(big), 1000 * k, 1000 * k + 2500))
    t = time.perf_counter()
    wp, tp = _windows(P, str(reads), str(paf), str(targets), "kC", 8, False, monkeypatch)
    ws, ts = _windows(P, str(reads), str(paf), str(targets), "kC", 8, True, monkeypatch)
    _same(ws, wp)
    assert wp.n_windows == 2 * (40_000_000 // 500) + 2 * 6
    assert tp < 20.0, tp          # (quadratic re-framing of a 40 MB record was ~10 rescans of up to 40 MB each: still seconds; the bound catches a regression on CI-sized boxes)


def test_reads_and_targets_in_one_file_are_parsed_once(P, tmp_path, monkeypatch):
    """`racon -f reads overlaps reads`: one file is both (same inode: a hard link counts, a copy does not).  The reads are then walked
    as the targets they duplicate instead of being parsed a second time (polisher.cpp initialize): the same windows as with the second
    parse (RACON_HIP_NO_SAME_FILE=1) and as with a byte-identical COPY of the file -- duplicate names included (reference
    src/polisher.cpp:223-278: the last target of a name is the one a read of that name maps to)."""
    import shutil
    reads = str(tmp_path / "reads.fastq")
    with gzip.open(DATA + "sample_reads.fastq.gz", "rb") as f:
        blob = f.read()
    second = blob.index(b"\n@2\n") + 1                    # (multi-line FASTQ: the first record ends where "@2" starts a line)
    open(reads, "wb").write(blob + blob[:second])         # the first record once more at the end: a duplicate name with equal data
    link, copy = str(tmp_path / "link.fastq"), str(tmp_path / "copy.fastq")
    os.link(reads, link); shutil.copy(reads, copy)
    ovl = str(tmp_path / "some.paf")                      # (every overlap costs a pairwise alignment: the first 400 of the 4 000 will do)
    with gzip.open(DATA + "sample_ava_overlaps.paf.gz", "rb") as f:
        open(ovl, "wb").write(b"".join(f.readlines()[:400]))
    for threads, serial in ((8, False), (1, True)):
        a, _ = _windows(P, reads, ovl, reads, "kF", threads, serial, monkeypatch)
        b, _ = _windows(P, reads, ovl, link, "kF", threads, serial, monkeypatch)
        c, _ = _windows(P, reads, ovl, copy, "kF", threads, serial, monkeypatch)
        monkeypatch.setenv("RACON_HIP_NO_SAME_FILE", "1")
        d, _ = _windows(P, reads, ovl, reads, "kF", threads, serial, monkeypatch)
        monkeypatch.delenv("RACON_HIP_NO_SAME_FILE")
        _same(a, b); _same(a, c); _same(a, d)
        assert a.n_windows > 236 and int(a.win_seq_off[-1]) > a.n_windows + 300
