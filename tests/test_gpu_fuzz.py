"""-m gpu: randomized windows built to stress the tie-break machinery rather than to look like data:
low-complexity backbones (many co-optimal alignments, many sinks with equal scores, weight ties in the
heaviest bundle), deep stacks of short partial layers (Subgraph path), layers longer than the backbone, zero
qualities (weight-0 edges: branch completion), IUPAC symbols, duplicate `begin` values (unstable std::sort
order), more than six in-edges per node.  HIP engine vs oracle, byte for byte."""
import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from helpers import assert_same

pytestmark = pytest.mark.gpu


def random_window(rng, style):
    alpha = [b"ACGT", b"AC", b"A", b"ACGTN", b"ACGTRYKM"][style % 5]
    L = int(rng.integers(8, 90))
    if style % 3 == 0:      # tandem repeat backbone
        unit = bytes(rng.choice(list(alpha), int(rng.integers(1, 4))).tolist())
        bb = (unit * (L // len(unit) + 1))[:L]
    else:
        bb = bytes(rng.choice(list(alpha), L).tolist())
    n_layers = int(rng.integers(0, 70))
    seqs = [(bb, bytes([33 + int(rng.integers(0, 20))]) * L if rng.random() < 0.5 else b"!" * L, 0, 0)]
    for _ in range(n_layers):
        full = rng.random() < 0.5
        b0 = 0 if full else int(rng.integers(0, max(1, L - 2)))
        e0 = L - 1 if full else int(rng.integers(b0 + 1, L))
        src = bytearray(bb[b0:e0 + 1])
        out = bytearray()
        for ch in src:                                     # noisy copy
            r = rng.random()
            if r < 0.08:
                continue
            out.append(int(rng.choice(list(alpha))) if r < 0.20 else ch)
            if rng.random() < 0.08:
                out.append(int(rng.choice(list(alpha))))
        if rng.random() < 0.1:
            out += bytes(rng.choice(list(alpha), int(rng.integers(1, 12))).tolist())      # overhanging tail
        if len(out) == 0:
            out = bytearray(b"A")
        s = bytes(out)
        q = None if rng.random() < 0.3 else bytes((rng.integers(0, 3, len(s)) * int(rng.integers(0, 15)) + 33).astype(np.uint8).tolist())
        seqs.append((s, q, b0, e0))
    return {"type": int(rng.integers(0, 2)), "seqs": seqs}


@pytest.mark.parametrize("seed,scores", [(1, (3, -5, -4)), (2, (5, -4, -8)), (3, (1, -1, -1)), (4, (3, -5, -4)), (5, (2, -3, -2))])
def test_fuzz_windows(oracle, seed, scores):
    from racon_amd.engine import HipEngine
    rng = np.random.default_rng(1000 + seed)
    wins = [random_window(rng, k) for k in range(400)]
    b = WindowBatch.from_windows(wins)
    ref = oracle.consensus(b, *scores, True, 0)
    got = HipEngine(*scores, True).consensus(b)
    assert_same(got, ref, f"fuzz seed {seed} scores {scores}")
    ref2 = oracle.consensus(b, *scores, False, 0)
    got2 = HipEngine(*scores, False).consensus(b)
    assert_same(got2, ref2, f"fuzz seed {seed} scores {scores} no trim")


@pytest.mark.parametrize("seed,scores", [(6, (3, -5, -4)), (7, (5, -4, -8)), (8, (1, -1, -1))])
def test_fuzz_windows_exact_order_consensus(oracle, seed, scores, monkeypatch):
    """Every window through the order-dependent consensus path: the parallel restatement of spoa's DFS
    TopologicalSort (phase_toposort4), the heaviest bundle over that order and BranchCompletion on it --
    normally taken only when the best node is not a sink or several nodes tie for the best score."""
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    monkeypatch.setenv("RCN_FORCE_EXACT", "1")
    rng = np.random.default_rng(2000 + seed)
    wins = [random_window(rng, k) for k in range(400)]
    b = WindowBatch.from_windows(wins)
    assert_same(HipEngine(*scores, True).consensus(b), oracle.consensus(b, *scores, True, 0), f"forced exact, fuzz seed {seed}")
    b2 = simulate_windows(60000, 500, 30.0, 10000, seed=300 + seed)          # ONT-like windows (deep graphs)
    assert_same(HipEngine(*scores, True).consensus(b2), oracle.consensus(b2, *scores, True, 0), f"forced exact, synthetic seed {seed}")
    b3 = simulate_windows(30000, 200, 60.0, 150, sub=0.003, ins=0.0005, dele=0.0005, seed=400 + seed,
                          phred_mean=30.0, phred_sd=0.0, phred_lo=30, phred_hi=30)    # short reads: Subgraph layers
    assert_same(HipEngine(*scores, True).consensus(b3), oracle.consensus(b3, *scores, True, 0), f"forced exact, short reads seed {seed}")


@pytest.mark.parametrize("level,seed,scores", [(3, 9, (3, -5, -4)), (3, 10, (1, -1, -1)), (2, 11, (3, -5, -4)), (2, 12, (5, -4, -8))])
def test_fuzz_windows_forced_sink_tie_levels(oracle, level, seed, scores, monkeypatch):
    """Sink ties (several sinks share the best NW score: spoa takes the first in ITS rank order) resolved by the later
    levels only: RCN_FORCE_TIE=3 sends every tie to phase_sink_tie_full (spoa's whole DFS TopologicalSort -- never needed
    on the ordinary test sets), RCN_FORCE_TIE=2 discards the id / backbone-position rule so that the bubble search plus the
    local DFS (phase_sink_tie_starts / phase_sink_tie_local) must reproduce its answers."""
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    monkeypatch.setenv("RCN_FORCE_TIE", str(level))
    rng = np.random.default_rng(3000 + seed)
    wins = [random_window(rng, k) for k in range(400)]
    b = WindowBatch.from_windows(wins)
    eng = HipEngine(*scores, True)
    assert_same(eng.consensus(b), oracle.consensus(b, *scores, True, 0), f"forced tie level {level}, fuzz seed {seed}")
    if level == 3:
        assert eng.stats()["n_sink_ties"] > 0, "the fuzz set must contain sink ties, else this test checks nothing"
    b3 = simulate_windows(20000, 200, 60.0, 150, sub=0.003, ins=0.0005, dele=0.0005, seed=500 + seed,
                          phred_mean=30.0, phred_sd=0.0, phred_lo=30, phred_hi=30)    # short reads: Subgraph layers, many ties
    eng3 = HipEngine(*scores, True)
    assert_same(eng3.consensus(b3), oracle.consensus(b3, *scores, True, 0), f"forced tie level {level}, short reads seed {seed}")


@pytest.mark.parametrize("seed,scores", [(13, (3, -5, -4)), (14, (5, -4, -8))])
def test_fuzz_windows_forced_slow_traceback(oracle, seed, scores, monkeypatch):
    """RCN_FORCE_SLOW_TB: every traceback step is the one-cell step against HBM (traceback2_slow_step: normally only taken
    when a predecessor lies beyond the staged tile or a node has more than six in-edges)."""
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    monkeypatch.setenv("RCN_FORCE_SLOW_TB", "1")
    rng = np.random.default_rng(4000 + seed)
    wins = [random_window(rng, k) for k in range(200)]
    b = WindowBatch.from_windows(wins)
    assert_same(HipEngine(*scores, True).consensus(b), oracle.consensus(b, *scores, True, 0), f"slow traceback, fuzz seed {seed}")
    b2 = simulate_windows(6000, 500, 12.0, 3000, seed=600 + seed)
    assert_same(HipEngine(*scores, True).consensus(b2), oracle.consensus(b2, *scores, True, 0), f"slow traceback, synthetic seed {seed}")


@pytest.mark.gpu
@pytest.mark.parametrize("kw,scores", [(dict(window_len=500, coverage=30.0, read_len=8000, low_complexity=0.9, n_rate=0.002), (3, -5, -4)),
                                        (dict(window_len=200, coverage=60.0, read_len=150, sub=0.003, ins=0.002, dele=0.002, low_complexity=0.6, n_rate=0.02), (5, -4, -8)),
                                        (dict(window_len=300, coverage=100.0, read_len=3000, sub=0.08, ins=0.06, dele=0.08, low_complexity=1.0), (1, -1, -1))])
def test_low_complexity_contigs(oracle, kw, scores):
    """Homopolymer runs, short tandem repeats, two-letter stretches, N bases (synth._low_complexity): alignments with many
    co-optimal paths -- ties in the predecessor choice, the move priority and between sinks -- through both kernels
    (tools/soak_vs_oracle.py --lowcomplexity is the same at size: profiles/r06/y_soak_low_complexity_vs_oracle.json)."""
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    b = simulate_windows(150_000, seed=31, **kw)
    for trim in (True, False):
        ref = oracle.consensus(b, *scores, trim, 0)
        got = HipEngine(*scores, trim).consensus(b)
        assert_same(got, ref, f"low-complexity {kw} scores {scores} trim {trim}")

