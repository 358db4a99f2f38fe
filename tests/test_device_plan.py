"""not gpu: the device-built path's plan and shard inputs (racon_amd/host/device_job.cpp: plan_device_job, make_shard_input) through
include/racon_host.h -- pure host code.  A job cut into window ranges must be the same job: the windows the construction
(oracle/window_layout.py, pinned in tests/test_window_layout.py) makes from every shard's input, taken over the shard's own range, are
the windows of the uncut job -- contig mode (one target: every shard holds it) and fragment mode (236 targets that are also the
reads: a shard's input holds its targets and, as plain reads, the other shards' targets its overlaps point into)."""
import os

import numpy as np
import pytest

from helpers import REFDATA as DATA
needs_data = pytest.mark.skipif(not os.path.isdir(DATA), reason="reference test data not present")

CASES = [("sample_reads.fastq.gz", "sample_overlaps.sam.gz", "sample_layout.fasta.gz", "kC", 500),
         ("sample_reads.fastq.gz", "sample_ava_overlaps.paf.gz", "sample_reads.fastq.gz", "kF", 500)]


def window_rows(b, w):
    """Everything of window w of a WindowBatch as comparable Python data."""
    s0, s1 = int(b.win_seq_off[w]), int(b.win_seq_off[w + 1])
    out = [int(b.win_type[w])]
    for s in range(s0, s1):
        a, z = int(b.seq_off[s]), int(b.seq_off[s + 1])
        out.append((b.bases[a:z].tobytes(), b.quals[a:z].tobytes() if b.seq_has_qual[s] else None, int(b.seq_begin[s]), int(b.seq_end[s])))
    return out


@needs_data
@pytest.mark.parametrize("reads,overlaps,targets,typ,w", CASES)
@pytest.mark.parametrize("n_shards", [1, 3, 7])
def test_shards_rebuild_the_job(reads, overlaps, targets, typ, w, n_shards):
    from oracle.window_layout import window_layout
    from racon_amd import polisher
    polisher.build()
    p = polisher.Polisher(DATA + reads, DATA + overlaps, DATA + targets, typ, w, 10.0, 0.3, True, 3, -5, -4, 4)
    p.initialize(keep_layout=True)
    whole = p.windows()
    nw = whole.n_windows
    r, o, wt, wl, qt = p.layout()
    plan = p.device_plan(n_shards)
    n = plan["n_shards"]
    cut = plan["cut"]
    assert n == min(n_shards, nw) and cut[0] == 0 and cut[n] == nw and (np.diff(cut) >= 0).all()
    # every overlap is in at least one shard, and the shards are balanced by the overlaps over their windows (within a factor of the ideal)
    assert plan["n_overlaps"].sum() >= o.n_overlaps and (n == 1 or plan["n_overlaps"].max() <= 2.5 * o.n_overlaps / n + 64)
    seen = 0
    for s in range(n):
        d, rs, os_ = p.shard_input(n_shards, s)
        assert (d["window_first"], d["window_last"]) == (cut[s], cut[s + 1])
        assert d["window_base"] <= cut[s] and d["window_base"] + d["n_windows_local"] >= cut[s + 1]
        assert rs.n_targets == plan["target_hi"][s] - plan["target_lo"][s] and os_.n_overlaps == plan["n_overlaps"][s]
        if n == 1:
            assert rs.n_seqs == r.n_seqs and os_.n_overlaps == o.n_overlaps
        elif typ == "kF":
            assert rs.n_seqs < r.n_seqs and rs.n_targets < r.n_targets           # its own targets and the reads it needs, not everybody's
        local = window_layout(rs, os_, wl, qt, wt)
        assert local.n_windows == d["n_windows_local"]
        for gw in range(cut[s], cut[s + 1]):
            assert window_rows(local, gw - d["window_base"]) == window_rows(whole, gw), (s, gw)
            seen += 1
    assert seen == nw
    p.close()


@needs_data
def test_plan_needs_a_layout_and_rejects_bad_shards():
    from racon_amd import polisher
    polisher.build()
    p = polisher.Polisher(DATA + "sample_reads.fastq.gz", DATA + "sample_overlaps.sam.gz", DATA + "sample_layout.fasta.gz", "kC", 500, 10.0, 0.3, True, 3, -5, -4, 2)
    p.initialize()
    with pytest.raises(Exception):
        p.device_plan(2)
    p.close()
    p = polisher.Polisher(DATA + "sample_reads.fastq.gz", DATA + "sample_overlaps.sam.gz", DATA + "sample_layout.fasta.gz", "kC", 500, 10.0, 0.3, True, 3, -5, -4, 2)
    p.initialize(keep_layout=True)
    assert p.device_plan(1000)["n_shards"] == 96            # never more shards than windows
    with pytest.raises(Exception):
        p.shard_input(3, 5)
    p.close()
