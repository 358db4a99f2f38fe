"""Window construction (reference src/polisher.cpp:388-461).

not gpu: oracle/window_layout.py — the CPU restatement that checks rcn_engine_build_windows — is pinned here against
         the windows the host layer (racon_amd/host, which reproduces all goldens of the reference's test suite) builds
         from the reference's own test data: SAM and PAF overlaps, FASTQ and FASTA reads, contig and fragment mode.
gpu    : tests/test_gpu_window_build.py compares the HIP path with this oracle.
"""
import os

import numpy as np
import pytest

from helpers import REFDATA as DATA
needs_data = pytest.mark.skipif(not os.path.isdir(DATA), reason="reference test data not present")

FIELDS = ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals")


def same_batch(a, b, tag=""):
    for f in FIELDS:
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape, f"{tag}: {f} shape {x.shape} != {y.shape}"
        if not (x == y).all():
            k = int(np.nonzero(x != y)[0][0])
            raise AssertionError(f"{tag}: {f} differs first at {k}: {x[k]} != {y[k]}")


@needs_data
@pytest.mark.parametrize("reads,overlaps,targets,typ,w,q", [
    ("sample_reads.fastq.gz", "sample_overlaps.sam.gz", "sample_layout.fasta.gz", "kC", 500, 10.0),
    ("sample_reads.fasta.gz", "sample_overlaps.paf.gz", "sample_layout.fasta.gz", "kC", 500, 10.0),
    ("sample_reads.fastq.gz", "sample_overlaps.paf.gz", "sample_layout.fasta.gz", "kC", 1000, 10.0),
    ("sample_reads.fastq.gz", "sample_overlaps.sam.gz", "sample_layout.fasta.gz", "kC", 200, 14.0),
    ("sample_reads.fastq.gz", "sample_ava_overlaps.paf.gz", "sample_reads.fastq.gz", "kF", 500, 10.0),
])
def test_oracle_layout_equals_host_layer_on_reference_data(reads, overlaps, targets, typ, w, q):
    from oracle.window_layout import window_layout
    from racon_amd import polisher
    polisher.build()
    p = polisher.Polisher(DATA + reads, DATA + overlaps, DATA + targets, typ, w, q, 0.3, True, 5, -4, -8, 2)
    p.initialize(keep_layout=True)
    host = p.windows()
    r, o, wt, wl, qt = p.layout()
    assert wl == w and qt == q
    mine = window_layout(r, o, wl, qt, wt)
    same_batch(mine, host, f"{reads} {overlaps} {typ} w={w}")
    assert (o.strand == 1).any() and (o.strand == 0).any()


def test_synthetic_layout_edge_cases():
    """The generator used by the GPU tests: both strands, reads without quality, the -q filter firing, the length filter."""
    from oracle.window_layout import window_layout
    from racon_amd.synth import simulate_layout
    r, o, wt = simulate_layout(contig_lens=(9000, 2501, 499), read_len=1500, coverage=12, seed=5)
    b = window_layout(r, o, 500, 10.0, wt)
    assert b.n_windows == 18 + 6 + 1
    n_pairs = int(o.bp_off[-1]) // 2
    n_layers = b.n_seqs - b.n_windows
    assert 0 < n_layers < n_pairs                       # some pairs are dropped by the filters
    assert (b.seq_has_qual == 0).any() and (o.strand == 1).any()
    # with the quality filter off more layers survive
    assert window_layout(r, o, 500, 0.0, wt).n_seqs > b.n_seqs


def test_layout_rejects_invalid_layer():
    from oracle.window_layout import LayoutError, window_layout
    from racon_amd.layout import OverlapSet, ReadSet
    r = ReadSet.from_sequences([(b"ACGT" * 50, None), (b"ACGT" * 20, None)], 1)
    o = OverlapSet.from_lists([(1, 0, 0, [(150, 0), (100, 60)])])      # end before begin: add_layer's fatal error
    with pytest.raises(LayoutError):
        window_layout(r, o, 500, 10.0, 0)


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_layout_fixture(name="layout_sam_fastq_w500.npz"):
    from racon_amd.layout import OverlapSet, ReadSet
    z = np.load(os.path.join(GOLD, name))
    r = ReadSet(int(z["n_targets"]), z["seq_off"], z["bases"], z["quals"], z["seq_has_qual"])
    o = OverlapSet(z["q_id"], z["t_id"], z["strand"], z["bp_off"], z["bp_t"], z["bp_q"])
    return r, o, int(z["window_type"]), int(z["window_length"]), float(z["quality_threshold"])


def test_oracle_layout_on_committed_fixture():
    """tests/golden/layout_sam_fastq_w500.npz (tools/make_golden_layout.py: the reference's sample reads + SAM overlaps as
    the host layer flattens them) -> the windows of tests/golden/sam_fastq_w500.npz (built by the host layer, golden 1317)."""
    from oracle.window_layout import window_layout
    from racon_amd.batch import WindowBatch
    r, o, wt, wl, qt = load_layout_fixture()
    same_batch(window_layout(r, o, wl, qt, wt), WindowBatch.load(os.path.join(GOLD, "sam_fastq_w500.npz")), "layout fixture")


@needs_data
@pytest.mark.parametrize("overlaps,w", [("sample_overlaps.sam.gz", 500), ("sample_overlaps.paf.gz", 500), ("sample_overlaps.sam.gz", 137)])
def test_oracle_breaking_points_equal_host_layer_on_reference_data(overlaps, w):
    """oracle.window_layout.breaking_points (reference src/overlap.cpp:226-292 restated) against the breaking points the host
    layer derives from the same alignments: the SAM file's CIGARs, and for PAF the host's own pairwise alignments."""
    from oracle.window_layout import breaking_points
    from racon_amd import polisher
    polisher.build()
    p = polisher.Polisher(DATA + "sample_reads.fastq.gz", DATA + overlaps, DATA + "sample_layout.fasta.gz", "kC", w, 10.0, 0.3, True, 5, -4, -8, 2)
    p.initialize(keep_layout=True)
    _, o, _, _, _ = p.layout()
    mine = breaking_points(p.alignments(), w)
    for f in ("q_id", "t_id", "strand", "bp_off", "bp_t", "bp_q"):
        assert (np.asarray(getattr(mine, f)) == np.asarray(getattr(o, f))).all(), f


def test_synthetic_cigars_give_the_simulators_breaking_points():
    from oracle.window_layout import breaking_points
    from racon_amd.synth import simulate_layout
    r, o, wt, al = simulate_layout(contig_lens=(7000, 1501), read_len=1200, coverage=10, seed=9, with_cigars=True, window_len=300)
    mine = breaking_points(al, 300)
    for f in ("q_id", "t_id", "strand", "bp_off", "bp_t", "bp_q"):
        assert (np.asarray(getattr(mine, f)) == np.asarray(getattr(o, f))).all(), f
