#!/usr/bin/env python
"""Generates tests/golden/layout_*.npz from the reference's own test data (run in the build container, where
/root/reference exists; the GPU box only sees the committed output).

For the SAM / FASTQ / w=500 case of the reference's golden tests (reference test/racon_test.cpp:133-154) it stores the
INPUT of the window construction (reference src/polisher.cpp:388-461) as the host layer flattens it — every sequence on
its forward strand, every kept overlap with its breaking points and with the alignment (CIGAR + extents) they came from
(include/racon_hip.h: rcn_read_set / rcn_overlap_set / rcn_cigar_set).
The expected OUTPUT is already committed: tests/golden/sam_fastq_w500.npz, the windows the host layer built from the
same data (tools/make_golden.py asserts the reference's golden edit distance 1317 on them).  This script re-checks that
oracle/window_layout.py maps one onto the other before writing anything.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle.window_layout import window_layout  # noqa: E402
from racon_amd import polisher as P  # noqa: E402
from racon_amd.batch import WindowBatch  # noqa: E402

D = "/root/reference/test/data/"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    p = P.Polisher(D + "sample_reads.fastq.gz", D + "sample_overlaps.sam.gz", D + "sample_layout.fasta.gz", "kC", 500, 10, 0.3,
                   True, 5, -4, -8, num_threads=8)
    p.initialize(keep_layout=True)
    r, o, wt, wl, qt = p.layout()
    want = WindowBatch.load(os.path.join(OUT, "sam_fastq_w500.npz"))
    got = window_layout(r, o, wl, qt, wt)
    for f in ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals"):
        assert (np.asarray(getattr(got, f)) == np.asarray(getattr(want, f))).all(), f
    # only the reads some overlap uses travel (the sample has reads without overlaps); ids are remapped
    used = np.zeros(r.n_seqs, bool)
    used[:r.n_targets] = True
    used[o.q_id] = True
    new_id = np.cumsum(used) - 1
    off = r.seq_off.astype(np.int64)
    keep = np.nonzero(used)[0]
    lens = (off[1:] - off[:-1])[keep]
    seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    idx = np.concatenate([np.arange(off[k], off[k + 1]) for k in keep])
    al = p.alignments()
    from oracle.window_layout import breaking_points
    bp = breaking_points(al, wl)
    assert (bp.bp_t == o.bp_t).all() and (bp.bp_q == o.bp_q).all() and (bp.bp_off == o.bp_off).all()
    path = os.path.join(OUT, "layout_sam_fastq_w500.npz")
    np.savez_compressed(path, n_targets=r.n_targets, seq_off=seq_off, bases=r.bases[idx], quals=r.quals[idx],
                        seq_has_qual=r.seq_has_qual[keep], q_id=new_id[o.q_id].astype(np.uint32), t_id=o.t_id, strand=o.strand,
                        bp_off=o.bp_off, bp_t=o.bp_t, bp_q=o.bp_q, q_start=al.q_start, t_begin=al.t_begin, t_end=al.t_end,
                        cigar_off=al.cigar_off, cigar=al.cigar, window_type=wt, window_length=wl, quality_threshold=qt)
    print(path, os.path.getsize(path) >> 10, "KiB;", len(keep), "sequences,", o.n_overlaps, "overlaps,", int(o.bp_off[-1]) // 2, "pairs")


if __name__ == "__main__":
    main()
