"""The reference's own test suite (reference test/racon_test.cpp:53-295), re-stated against this
repo's host layer (racon_amd/host: createPolisher / initialize / windows / assemble) with the CPU
ORACLE as the consensus backend.  This is what pins the oracle: every golden number of the
reference's CPU tests must come out exactly.  Needs /root/reference/test/data (build container
only; skipped on the GPU box, where tests/golden/ carries the derived fixtures instead).

-m gpu twins at the bottom run the same pipelines with the HIP engine (racon::Polisher::polish
on the MI355X) where the reference data is present.
"""
import gzip
import os
import re

import pytest

DATA = "/root/reference/test/data/"
pytestmark = pytest.mark.skipif(not os.path.isdir(DATA), reason="reference test data not present")


@pytest.fixture(scope="module")
def P():
    from racon_amd import polisher
    polisher.build()
    return polisher


@pytest.fixture(scope="module")
def reference_contig():
    return b"".join(gzip.open(DATA + "sample_reference.fasta.gz").read().split(b"\n")[1:])


def revcomp(s: bytes) -> bytes:
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


# ---- RaconInitializeTest (racon_test.cpp:53-84): createPolisher's fatal errors --------------------
FASTX = (r"file  has unsupported format extension .valid extensions: .fasta, .fasta.gz, .fna, .fna.gz, "
         r".fa, .fa.gz, .fastq, .fastq.gz, .fq, .fq.gz.!")


@pytest.mark.parametrize("args,message", [
    (("", "", "", 3, 0), r".racon::createPolisher. error: invalid polisher type!"),                         # :53-57
    (("", "", "", "kC", 0), r".racon::createPolisher. error: invalid window length!"),                      # :59-62
    (("", "", "", "kC", 500), r".racon::createPolisher. error: " + FASTX),                                  # :64-69
    ((DATA + "sample_reads.fastq.gz", "", "", "kC", 500),
     r".racon::createPolisher. error: file  has unsupported format extension .valid extensions: .mhap, "
     r".mhap.gz, .paf, .paf.gz, .sam, .sam.gz.!"),                                                          # :71-76
    ((DATA + "sample_reads.fastq.gz", DATA + "sample_overlaps.paf.gz", "", "kC", 500),
     r".racon::createPolisher. error: " + FASTX),                                                           # :78-84
])
def test_create_polisher_errors(P, args, message):
    with pytest.raises(P.RaconError) as e:
        P.Polisher(args[0], args[1], args[2], args[3], args[4], 0, 0, False, 0, 0, 0, 0)
    assert re.search(message, str(e.value)), str(e.value)


# ---- RaconPolishingTest, contig polishing (racon_test.cpp:86-222) ------------------------------------
CONTIG = [  # reads, overlaps, window, scores, golden edit distance, line of the EXPECT_EQ
    ("sample_reads.fastq.gz", "sample_overlaps.paf.gz", 500, (5, -4, -8), 1312, 104),
    ("sample_reads.fasta.gz", "sample_overlaps.paf.gz", 500, (5, -4, -8), 1566, 128),
    ("sample_reads.fastq.gz", "sample_overlaps.sam.gz", 500, (5, -4, -8), 1317, 151),
    ("sample_reads.fasta.gz", "sample_overlaps.sam.gz", 500, (5, -4, -8), 1770, 174),
    ("sample_reads.fastq.gz", "sample_overlaps.paf.gz", 1000, (5, -4, -8), 1289, 197),
    ("sample_reads.fastq.gz", "sample_overlaps.paf.gz", 500, (1, -1, -1), 1321, 220),
]


def _contig_case(P, backend, reads, ovl, w, sc):
    p = P.Polisher(DATA + reads, DATA + ovl, DATA + "sample_layout.fasta.gz", "kC", w, 10, 0.3, True, *sc, num_threads=4)
    p.initialize()
    if backend == "hip":
        fa = P.parse_fasta(p.polish(True))
    else:
        res = backend.consensus(p.windows(), *sc, True, 0)
        fa = P.parse_fasta(p.assemble(res, True))
    assert len(fa) == 1
    return fa[0][1]


@pytest.mark.parametrize("reads,ovl,w,sc,gold,line", CONTIG)
def test_contig_goldens_oracle(P, oracle, reference_contig, reads, ovl, w, sc, gold, line):
    seq = _contig_case(P, oracle, reads, ovl, w, sc)
    assert P.edit_distance(revcomp(seq), reference_contig) == gold, f"racon_test.cpp:{line}"


# ---- RaconPolishingTest, fragment correction (racon_test.cpp:224-295) ------------------------------
FRAGMENT = [  # reads (= targets), overlaps, type, drop unpolished, (#sequences, total length), lines
    ("sample_reads.fastq.gz", "sample_ava_overlaps.paf.gz", "kC", True, (40, 401246), "234,240"),
    ("sample_reads.fastq.gz", "sample_ava_overlaps.paf.gz", "kF", False, (236, 1658216), "252,258"),
    ("sample_reads.fasta.gz", "sample_ava_overlaps.paf.gz", "kF", False, (236, 1663982), "270,276"),
    ("sample_reads.fastq.gz", "sample_ava_overlaps.mhap.gz", "kF", False, (236, 1658216), "288,294"),
]


def _fragment_case(P, backend, reads, ovl, ty, drop):
    p = P.Polisher(DATA + reads, DATA + ovl, DATA + reads, ty, 500, 10, 0.3, True, 1, -1, -1, num_threads=8)
    p.initialize()
    if backend == "hip":
        return P.parse_fasta(p.polish(drop))
    res = backend.consensus(p.windows(), 1, -1, -1, True, 0)
    return P.parse_fasta(p.assemble(res, drop))


@pytest.mark.parametrize("reads,ovl,ty,drop,gold,lines", FRAGMENT)
def test_fragment_goldens_oracle(P, oracle, reads, ovl, ty, drop, gold, lines):
    fa = _fragment_case(P, oracle, reads, ovl, ty, drop)
    assert (len(fa), sum(len(s) for _, s in fa)) == gold, f"racon_test.cpp:{lines}"


# ---- the same pipelines end-to-end on the MI355X (only where the reference data exists) ---------------
@pytest.mark.gpu
@pytest.mark.parametrize("reads,ovl,w,sc,gold,line", CONTIG)
def test_contig_goldens_hip(P, reference_contig, reads, ovl, w, sc, gold, line):
    seq = _contig_case(P, "hip", reads, ovl, w, sc)
    assert P.edit_distance(revcomp(seq), reference_contig) == gold, f"racon_test.cpp:{line}"


@pytest.mark.gpu
@pytest.mark.parametrize("reads,ovl,ty,drop,gold,lines", FRAGMENT[:2])
def test_fragment_goldens_hip(P, reads, ovl, ty, drop, gold, lines):
    fa = _fragment_case(P, "hip", reads, ovl, ty, drop)
    assert (len(fa), sum(len(s) for _, s in fa)) == gold, f"racon_test.cpp:{lines}"
