"""-m gpu: the exact banded DP (racon_amd/csrc/poa_band.hpp).  Layers of 256..639 bases are aligned over a 256-column
window per row; a certificate decides per alignment whether the result is provably the full matrix's, else the alignment
is redone on full rows.  Either way the consensus must equal the oracle's (which always evaluates the full matrix), on
ordinary data (certificate holds almost always), with the certificate forced to fail (redo path), on long pathological
windows (repeats, junk layers, overhangs: the certificate must fail where the band is wrong), and with banding off."""
import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_windows
from helpers import assert_same

pytestmark = pytest.mark.gpu


def long_window(rng, style):
    """A window whose layers are long enough for the banded pass, with structure that pulls optimal paths off the diagonal."""
    alpha = [b"ACGT", b"AC", b"ACGT", b"ACGTN"][style % 4]
    L = int(rng.integers(280, 620))
    if style % 3 == 0:      # tandem repeats: many co-optimal alignments far from the diagonal
        unit = bytes(rng.choice(list(alpha), int(rng.integers(2, 12))).tolist())
        bb = (unit * (L // len(unit) + 1))[:L]
    else:
        bb = bytes(rng.choice(list(alpha), L).tolist())
    seqs = [(bb, b"!" * L, 0, 0)]
    for k in range(int(rng.integers(3, 14))):
        full = rng.random() < 0.7
        b0 = 0 if full else int(rng.integers(0, L // 3))
        e0 = L - 1 if full else int(rng.integers(2 * L // 3, L))
        src = bytearray(bb[b0:e0 + 1])
        kind = rng.random()
        if kind < 0.15:       # a long deletion
            a = int(rng.integers(0, max(1, len(src) - 120))); del src[a:a + int(rng.integers(20, 120))]
        elif kind < 0.30:     # a long insertion
            a = int(rng.integers(0, len(src))); src[a:a] = bytes(rng.choice(list(alpha), int(rng.integers(20, 120))).tolist())
        elif kind < 0.36:     # unrelated sequence
            src = bytearray(rng.choice(list(alpha), len(src)).tolist())
        rate = [0.02, 0.1, 0.25][int(rng.integers(0, 3))]
        out = bytearray()
        for ch in src:
            r = rng.random()
            if r < rate / 3:
                continue
            out.append(int(rng.choice(list(alpha))) if r < rate else ch)
            if rng.random() < rate / 3:
                out.append(int(rng.choice(list(alpha))))
        if len(out) < 2:
            out = bytearray(b"AC")
        s = bytes(out[:639])
        q = None if rng.random() < 0.3 else bytes((rng.integers(0, 25, len(s)) + 33).astype(np.uint8).tolist())
        seqs.append((s, q, b0, e0))
    return {"type": int(rng.integers(0, 2)), "seqs": seqs}


def test_band_is_used_and_certified_on_ont_like_windows(oracle):
    from racon_amd.engine import HipEngine
    b = simulate_windows(150_000, 500, 30.0, 10000, seed=7001)
    eng = HipEngine(3, -5, -4, True)
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 0), "banded, cfg2-like")
    st = eng.stats()
    assert st["n_banded"] > 5000 and st["n_band_redone"] < 0.08 * st["n_banded"], (st["n_banded"], st["n_band_redone"], st["band_redo_why"])
    assert st["dp_cells"] < 0.62 * st["dp_cells_full"]            # about half of every banded row is not evaluated


@pytest.mark.parametrize("scores", [(3, -5, -4), (5, -4, -8), (1, -1, -1)])
def test_band_redo_path(oracle, scores, monkeypatch):
    """RCN_FORCE_BAND_FAIL: the banded pass runs, its certificate is discarded, every alignment is redone on full rows."""
    from racon_amd.engine import HipEngine
    monkeypatch.setenv("RCN_FORCE_BAND_FAIL", "1")
    b = simulate_windows(40_000, 500, 25.0, 10000, seed=7002)
    eng = HipEngine(*scores, True)
    assert_same(eng.consensus(b), oracle.consensus(b, *scores, True, 0), f"forced band redo {scores}")
    st = eng.stats()
    assert st["n_banded"] == 0 and st["n_band_redone"] > 1000 and st["dp_cells"] == st["dp_cells_full"]


def test_band_off_switch(oracle, monkeypatch):
    from racon_amd.engine import HipEngine
    monkeypatch.setenv("RCN_NO_BAND", "1")
    b = simulate_windows(40_000, 500, 25.0, 10000, seed=7003)
    eng = HipEngine(3, -5, -4, True)
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 0), "banding off")
    st = eng.stats()
    assert st["n_banded"] == 0 and st["n_band_redone"] == 0


@pytest.mark.parametrize("seed,scores", [(1, (3, -5, -4)), (2, (5, -4, -8)), (3, (1, -1, -1)), (4, (2, -3, -2))])
def test_band_on_long_pathological_windows(oracle, seed, scores):
    from racon_amd.engine import HipEngine
    rng = np.random.default_rng(7100 + seed)
    b = WindowBatch.from_windows([long_window(rng, k) for k in range(160)])
    eng = HipEngine(*scores, True)
    assert_same(eng.consensus(b), oracle.consensus(b, *scores, True, 0), f"long fuzz seed {seed} scores {scores}")
    st = eng.stats()
    assert st["n_banded"] + st["n_band_redone"] > 500


@pytest.mark.parametrize("w,seed", [(300, 7201), (400, 7202), (600, 7203)])
def test_band_window_lengths(oracle, w, seed):
    """Window lengths around the banded range: 300 (window barely narrower than the row), 400, 600 (rows of up to ~640)."""
    from racon_amd.engine import HipEngine
    b = simulate_windows(60_000, w, 25.0, 10000, seed=seed)
    eng = HipEngine(3, -5, -4, True)
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 0), f"w {w}")
    assert eng.stats()["n_banded"] > 0


@pytest.mark.parametrize("seed,scores", [(5, (3, -5, -4)), (6, (5, -4, -8))])
def test_band_that_keeps_the_scores(oracle, seed, scores, monkeypatch):
    """RCN_BAND_SCORES: the banded pass stores the int16 scores of its window (and guard cells) instead of move codes, and the
    score-reading traceback walks them -- the variant the move codes replaced, kept selectable."""
    from racon_amd.engine import HipEngine
    monkeypatch.setenv("RCN_BAND_SCORES", "1")
    rng = np.random.default_rng(7300 + seed)
    b = WindowBatch.from_windows([long_window(rng, k) for k in range(120)])
    assert_same(HipEngine(*scores, True).consensus(b), oracle.consensus(b, *scores, True, 0), f"band with scores, long fuzz {seed}")
    b2 = simulate_windows(40_000, 500, 25.0, 10000, seed=7300 + seed)
    eng = HipEngine(*scores, True)
    assert_same(eng.consensus(b2), oracle.consensus(b2, *scores, True, 0), f"band with scores, synthetic {seed}")
    assert eng.stats()["n_banded"] > 1000


@pytest.mark.parametrize("level", [2, 3])
def test_band_with_forced_sink_tie_levels(oracle, level, monkeypatch):
    """Sink ties after a coded alignment: the full-DFS level compares the end scores the DP kept aside (there is no score
    matrix to read them from)."""
    from racon_amd.engine import HipEngine
    monkeypatch.setenv("RCN_FORCE_TIE", str(level))
    rng = np.random.default_rng(7400 + level)
    b = WindowBatch.from_windows([long_window(rng, k) for k in range(120)])
    eng = HipEngine(3, -5, -4, True)
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 0), f"coded band, forced tie level {level}")


def test_sink_tie_through_the_closure_sweep_before_a_coded_traceback(oracle):
    """Round 6, found by the all-records check of cfg5 whole (one window in 1.34 M): several sinks tie, the rule does not decide, the tie
    goes through the closure sweep -- which borrows the descriptor array and has it rebuilt -- and the rebuild wrote 'row 0 of Z' into a
    matrix that held move codes (code rows 0 and 1); a traceback that reached row 1 with an insertion then went wrong.  The window it
    happened in (tests/golden/tie_closure_sweep_then_coded_traceback.npz, window 1) and 150 variations of it: layers dropped, reordered,
    bases put in front of some (paths that start with an insertion)."""
    import os
    from racon_amd.engine import HipEngine
    b0 = WindowBatch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tie_closure_sweep_then_coded_traceback.npz"))
    W = b0.window(1)
    rng = np.random.default_rng(7600)
    wins = [W]
    for k in range(150):
        seqs = list(W["seqs"][1:])
        keep = sorted(rng.choice(len(seqs), int(rng.integers(len(seqs) - 4, len(seqs) + 1)), replace=False).tolist())
        if k % 3 == 0:
            rng.shuffle(keep)
        lay = []
        for i in keep:
            s, q, bg, en = seqs[i]
            if rng.random() < 0.3:
                n = int(rng.integers(1, 3))
                s = bytes(rng.choice(list(b"ACGT"), n).tolist()) + s
                q = None if q is None else bytes([int(q[0])] * n) + q
            lay.append((s, q, bg, en))
        wins.append({"type": W["type"], "seqs": [W["seqs"][0]] + lay})
    b = WindowBatch.from_windows(wins)
    eng = HipEngine(3, -5, -4, True)
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 0), "tie through the closure sweep, coded traceback")
    assert eng.stats()["n_banded"] > 1000             # (a build with the old rebuild fails five of these windows and the fixture: profiles/r06/v_regression_old_and_fixed.txt)


def fan_in_window(rng, n_variants):
    """A deep window in which the node after every hot spot collects `n_variants` + 1 in-edges (backbone edge + one per
    local variant: inserted bases, deletions of 1..3 bases, substitutions of the base in front): rows with seven and
    eight predecessors, which the move codes name from the in-edge list, and with more, which fall back to full rows."""
    L = 420
    bb = bytearray(rng.choice(list(b"ACGT"), L).tolist())
    spots = [60, 150, 240, 330]
    for p in spots:                       # fixed context so that every variant is a distinct, unambiguous edit
        bb[p - 4:p + 2] = b"ACGTCA"
    bb = bytes(bb)
    variants = [("ins", b"G"), ("del", 1), ("sub", b"A"), ("ins", b"TT"), ("del", 2), ("sub", b"C"), ("ins", b"A"), ("del", 3),
                ("sub", b"G"), ("ins", b"CC")][:n_variants]
    seqs = [(bb, b"!" * L, 0, 0)]
    for k in range(4 * len(variants) + 6):
        s = bytearray(bb)
        if k < 4 * len(variants):
            kind, arg = variants[k % len(variants)]
            for p in reversed(spots):
                if kind == "ins": s[p:p] = arg
                elif kind == "del": del s[p - arg:p]
                else: s[p - 1:p] = arg
        s = bytes(s)
        seqs.append((s, bytes([33 + int(rng.integers(5, 40))]) * len(s), 0, L - 1))
    return {"type": 1, "seqs": seqs}


@pytest.mark.parametrize("n_variants,scores", [(8, (3, -5, -4)), (9, (3, -5, -4)), (9, (5, -4, -8)), (10, (3, -5, -4))])
def test_band_rows_with_seven_and_eight_in_edges(oracle, n_variants, scores):
    from racon_amd.engine import HipEngine
    rng = np.random.default_rng(7500 + n_variants)
    b = WindowBatch.from_windows([fan_in_window(rng, n_variants) for _ in range(24)])
    eng = HipEngine(*scores, True)
    assert_same(eng.consensus(b), oracle.consensus(b, *scores, True, 0), f"fan-in {n_variants} {scores}")
    st = eng.stats()
    assert st["n_banded"] > 0
    # (tests/emul with RCN_EMUL_VERBOSE counts them: 8 / 9 variants -> widest row has 7-8 in-edges in most alignments, 10 -> more)
    if scores != (3, -5, -4):
        return                            # other scores align the variants differently: in-edge counts not checked
    if n_variants <= 9:                   # at most eight in-edges: no alignment is sent back for that reason
        assert st["band_redo_why"][3] == 0, st["band_redo_why"]
    else:
        assert st["band_redo_why"][3] > 0, st["band_redo_why"]


@pytest.mark.parametrize("scores", [(1, 2, -1), (-2, -3, -1), (0, -1, -1), (2, 2, -2), (4, -6, 0), (3, -5, 2)])
def test_score_sets_the_band_certificate_does_not_cover(oracle, scores):
    """Any -m / -x / -g is legal (reference src/main.cpp:51-53,91-99).  The certificate bounds what a remaining base can
    add by m: sets with x > m or g > m must take full rows, g >= 0 the int32 kernel -- same bytes as the oracle either way."""
    from racon_amd.engine import HipEngine
    rng = np.random.default_rng(77)
    b = WindowBatch.from_windows([long_window(rng, s) for s in range(8)]).concat(simulate_windows(6000, 500, 12.0, 3000, seed=7009))
    eng = HipEngine(*scores, True)
    assert_same(eng.consensus(b), oracle.consensus(b, *scores, True, 0), f"scores {scores}")
    m, x, g = scores
    if x > m or g > m:
        assert eng.stats()["n_banded"] == 0
