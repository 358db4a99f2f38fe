"""-m gpu: rcn_engine_build_windows (windows built in HBM from resident reads + breaking points, reference
src/polisher.cpp:388-461) against oracle/window_layout.py, array for array, and the consensus of the built batch
against the consensus of the same batch uploaded from the host."""
import os

import numpy as np
import pytest

from test_window_layout import DATA, GOLD, load_layout_fixture, same_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,w,q", [
    (dict(contig_lens=(30000, 12345), seed=11), 500, 10.0),
    (dict(contig_lens=(9000, 2501, 499, 1), read_len=1500, coverage=12, seed=5), 500, 10.0),
    (dict(contig_lens=(20000,), read_len=3000, coverage=25, seed=6, target_quality=True), 1000, 8.0),
    (dict(contig_lens=(6000, 6000), read_len=150, coverage=40, seed=7, sub=0.003, ins=0.0005, dele=0.0005,
          frac_no_quality=0.0, frac_low_quality=0.0), 200, 10.0),
    (dict(contig_lens=(15000,), seed=8, frac_no_quality=1.0), 500, 10.0),
])
def test_build_windows_equals_oracle(oracle, kw, w, q):
    from oracle.window_layout import window_layout
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_layout
    r, o, wt = simulate_layout(window_len=w, **kw)
    ref = window_layout(r, o, w, q, wt)
    eng = HipEngine(3, -5, -4, True)
    eng.build_windows(r, o, w, q, wt)
    got = eng.export_batch()
    same_batch(got, ref, f"build_windows {kw}")
    st = eng.build_stats()
    assert st["n_pairs"] == int(o.bp_off[-1]) // 2 and st["n_layers"] == ref.n_seqs - ref.n_windows
    # the built batch polishes exactly like the uploaded one
    built = eng.run()
    up = HipEngine(3, -5, -4, True).consensus(ref)
    assert built.consensus == up.consensus and (np.asarray(built.polished) == np.asarray(up.polished)).all()
    cpu = oracle.consensus(ref, 3, -5, -4, True, 0)
    assert built.consensus == cpu.consensus


def test_build_windows_rejects_invalid_layer():
    from racon_amd.engine import HipEngine
    from racon_amd.layout import OverlapSet, ReadSet
    r = ReadSet.from_sequences([(b"ACGT" * 50, None), (b"ACGT" * 20, None)], 1)
    o = OverlapSet.from_lists([(1, 0, 0, [(150, 0), (100, 60)])])
    with pytest.raises(RuntimeError):
        HipEngine().build_windows(r, o, 500, 10.0, 0)


def test_build_windows_without_overlaps_gives_backbones():
    from racon_amd.engine import HipEngine
    from racon_amd.layout import OverlapSet, ReadSet
    r = ReadSet.from_sequences([(b"ACGT" * 300, None)], 1)
    o = OverlapSet.from_lists([])
    eng = HipEngine()
    eng.build_windows(r, o, 500, 10.0, 1)
    b = eng.export_batch()
    assert b.n_windows == 3 and b.n_seqs == 3 and bytes(b.bases) == b"ACGT" * 300
    res = eng.run()
    assert b"".join(res.consensus) == b"ACGT" * 300 and not any(res.polished)


@pytest.mark.skipif(not os.path.isdir(DATA), reason="reference test data not present")
def test_build_windows_on_reference_data(oracle):
    from racon_amd import polisher
    from racon_amd.engine import HipEngine
    polisher.build()
    p = polisher.Polisher(DATA + "sample_reads.fastq.gz", DATA + "sample_overlaps.sam.gz", DATA + "sample_layout.fasta.gz",
                          "kC", 500, 10.0, 0.3, True, 5, -4, -8, 2)
    p.initialize(keep_layout=True)
    r, o, wt, wl, qt = p.layout()
    eng = HipEngine(5, -4, -8, True)
    eng.build_windows(r, o, wl, qt, wt)
    same_batch(eng.export_batch(), p.windows(), "reference sample data")


def test_build_windows_on_committed_reference_fixture():
    """The reference's sample reads + SAM overlaps (tests/golden/layout_sam_fastq_w500.npz) -> exactly the windows the host
    layer builds from them (tests/golden/sam_fastq_w500.npz, golden edit distance 1317), and the same consensus."""
    from racon_amd.batch import WindowBatch
    from racon_amd.engine import HipEngine
    r, o, wt, wl, qt = load_layout_fixture()
    want = WindowBatch.load(os.path.join(GOLD, "sam_fastq_w500.npz"))
    eng = HipEngine(5, -4, -8, True)
    eng.build_windows(r, o, wl, qt, wt)
    same_batch(eng.export_batch(), want, "reference fixture")
    assert eng.run().consensus == HipEngine(5, -4, -8, True).consensus(want).consensus


@pytest.mark.parametrize("kw,w", [
    (dict(contig_lens=(30000, 12345), seed=21), 500),
    (dict(contig_lens=(9000, 2501, 499), read_len=1500, coverage=12, seed=22), 137),
    (dict(contig_lens=(20000,), read_len=3000, coverage=25, seed=23, ins=0.10, dele=0.12), 1000),
])
def test_build_windows_from_cigars_equals_oracle(oracle, kw, w):
    """Breaking points (reference src/overlap.cpp:226-292) on the device too: alignments in, the same batch out."""
    from oracle.window_layout import breaking_points, window_layout
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_layout
    r, o, wt, al = simulate_layout(window_len=w, with_cigars=True, **kw)
    ref = window_layout(r, breaking_points(al, w), w, 10.0, wt)
    eng = HipEngine(3, -5, -4, True)
    eng.build_windows_from_cigars(r, al, w, 10.0, wt)
    same_batch(eng.export_batch(), ref, f"from cigars {kw}")
    assert eng.run().consensus == oracle.consensus(ref, 3, -5, -4, True, 0).consensus


def test_build_windows_from_cigars_on_committed_reference_fixture():
    """The reference sample's SAM alignments (tests/golden/layout_sam_fastq_w500.npz) -> the windows of sam_fastq_w500.npz."""
    from racon_amd.batch import WindowBatch
    from racon_amd.engine import HipEngine
    from racon_amd.layout import CigarSet
    r, o, wt, wl, qt = load_layout_fixture()
    z = np.load(os.path.join(GOLD, "layout_sam_fastq_w500.npz"))
    al = CigarSet(o.q_id, o.t_id, o.strand, z["q_start"], z["t_begin"], z["t_end"], z["cigar_off"], z["cigar"])
    eng = HipEngine(5, -4, -8, True)
    eng.build_windows_from_cigars(r, al, wl, qt, wt)
    same_batch(eng.export_batch(), WindowBatch.load(os.path.join(GOLD, "sam_fastq_w500.npz")), "reference fixture, from CIGARs")


@pytest.mark.parametrize("serial", [False, True])
def test_cigar_walk_edge_cases(oracle, serial, monkeypatch):
    """Hand-made alignments for both device kernels of the CIGAR walk (wave per overlap; RCN_CIGAR_SERIAL=1: thread per
    overlap) against the base-by-base restatement of reference src/overlap.cpp:226-292: =/X/N operations, clips, zero
    counts, runs crossing many windows, deletions swallowing whole windows, numbers that straddle the 64-byte steps of the
    wave kernel, and a CIGAR that ends before the overlap's target extent (the last window is then never closed)."""
    from oracle.window_layout import breaking_points, window_layout
    from racon_amd.engine import HipEngine
    from racon_amd.layout import CigarSet, ReadSet
    if serial:
        monkeypatch.setenv("RCN_CIGAR_SERIAL", "1")
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    target = acgt[rng.integers(0, 4, 5000)].tobytes()
    reads = [(target, None)]
    al = []

    def add(cigar: bytes, t_begin: int, strand: int = 0, t_short: int = 0):
        import re
        ops = [(int(n or 0), op) for n, op in re.findall(rb"(\d*)([A-Z=])", cigar)]
        qlen = sum(n for n, op in ops if op in b"M=XI")
        tlen = sum(n for n, op in ops if op in b"M=XDN")
        clip = sum(n for n, op in ops if op in b"S")
        seq = acgt[rng.integers(0, 4, qlen + clip + 7)].tobytes()
        reads.append((seq, bytes((rng.integers(10, 30, len(seq)) + 33).astype(np.uint8).tolist())))
        al.append((len(reads) - 1, 0, strand, clip, t_begin, t_begin + tlen + t_short, cigar))

    add(b"10S5M2I3D100M5H", 37)
    add(b"2000M", 400)
    add(b"100M600D100M", 250, strand=1)
    add(b"50=3X0M20=1000N30X10=", 1200)
    add(b"1M" * 40 + b"1234M" + b"7I" * 3 + b"3D2M", 100)            # "1234M" straddles byte 64 of the text
    add(b"9M" * 31 + b"0I" + b"321M5D4M", 77)                            # the number ends exactly at the step border
    add(b"300M", 4600, t_short=60)                                       # CIGAR stops 60 columns before t_end
    add(b"499M", 1)
    add(b"1M498D1M", 2500)
    r = ReadSet.from_sequences(reads, 1)
    a = CigarSet.from_lists(al)
    for w in (500, 97):
        ref = window_layout(r, breaking_points(a, w), w, 0.0, 1)
        eng = HipEngine()
        eng.build_windows_from_cigars(r, a, w, 0.0, 1)
        same_batch(eng.export_batch(), ref, f"cigar edge cases w={w} serial={serial}")
