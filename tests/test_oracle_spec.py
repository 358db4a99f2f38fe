"""Oracle (oracle/poa_oracle.cpp) against the independent pure-Python executable
specification (oracle/spec_py.py) on small cases."""
import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_windows
from oracle import spec_py
from helpers import edge_case_batch, edge_case_windows


def spec_consensus(win, m, x, g, trim=True):
    seqs = [(s.decode("latin1"), (qq.decode("latin1") if qq is not None else None), b, e) for (s, qq, b, e) in win["seqs"]]
    c, p = spec_py.window_consensus({"seqs": seqs}, m, x, g, tgs=(win["type"] == 1), trim=trim)
    return c.encode("latin1"), p


@pytest.mark.parametrize("scores", [(3, -5, -4), (5, -4, -8), (1, -1, -1)])
def test_oracle_matches_spec_on_edge_cases(oracle, scores, capsys):
    b = edge_case_batch()
    r = oracle.consensus(b, *scores, True, 2)
    for w, win in enumerate(edge_case_windows()):
        c, p = spec_consensus(win, *scores)
        assert c == r.consensus[w], f"window {w}"
        assert p == bool(r.polished[w])


def test_oracle_matches_spec_on_synthetic(oracle):
    b = simulate_windows(3000, 500, 12, 2000, seed=5)
    r = oracle.consensus(b, 3, -5, -4, True, 2)
    for w in range(b.n_windows):
        c, p = spec_consensus(b.window(w), 3, -5, -4)
        assert c == r.consensus[w]
        assert p == bool(r.polished[w])


def test_low_complexity_contigs(oracle):
    """Homopolymer runs, short tandem repeats, two-letter stretches and N bases (synth._low_complexity): many co-optimal
    alignments, i.e. ties at every level.  The option leaves the seeded workloads without it unchanged; oracle = spec,
    and the oracle's AVX2 variant = its scalar one."""
    plain, again = simulate_windows(3000, 500, 12, 2000, seed=5), simulate_windows(3000, 500, 12, 2000, seed=5, low_complexity=0.0)
    assert plain.bases.tobytes() == again.bases.tobytes() and plain.seq_off.tobytes() == again.seq_off.tobytes()
    b = simulate_windows(2500, 250, 10, 1500, seed=11, low_complexity=0.9, n_rate=0.01)
    assert b.bases.tobytes() != simulate_windows(2500, 250, 10, 1500, seed=11).bases.tobytes() and ord("N") in b.bases
    r = oracle.consensus(b, 3, -5, -4, True, 2)
    for w in range(b.n_windows):
        c, p = spec_consensus(b.window(w), 3, -5, -4)
        assert c == r.consensus[w], f"window {w}"
        assert p == bool(r.polished[w])
    big = simulate_windows(40000, 500, 30, 8000, seed=12, low_complexity=0.8, n_rate=0.002)
    a, v = oracle.consensus(big, 3, -5, -4, True, 0, simd=False), oracle.consensus(big, 3, -5, -4, True, 0, simd=True)
    assert a.consensus == v.consensus


def test_oracle_trim_and_type_semantics(oracle):
    b = edge_case_batch()
    r_trim = oracle.consensus(b, 3, -5, -4, True, 1)
    r_no = oracle.consensus(b, 3, -5, -4, False, 1)
    bb = edge_case_windows()[0]["seqs"][0][0]
    assert r_trim.consensus[0] == bb and not r_trim.polished[0]          # < 3 sequences: backbone copy (window.cpp:68-71)
    assert r_trim.consensus[1] == bb and not r_trim.polished[1]
    assert r_trim.consensus[2] == bb and r_trim.polished[2]
    assert r_trim.consensus[5] == r_no.consensus[5]                        # kNGS: never trimmed (window.cpp:125)
    assert len(r_trim.consensus[6]) < len(r_no.consensus[6])               # kTGS + trim: ends stripped
    assert r_trim.chimeric[9] == 1 and r_no.chimeric[9] == 0              # warning only on the trim path
    assert r_trim.consensus[9] == r_no.consensus[9]                        # ... and the consensus is kept untrimmed


def test_edit_distance_helper(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = bytes(rng.integers(65, 69, rng.integers(1, 200)).astype(np.uint8))
        b = bytes(rng.integers(65, 69, rng.integers(1, 200)).astype(np.uint8))
        # reference DP
        D = list(range(len(b) + 1))
        for i in range(1, len(a) + 1):
            prev, D[0] = D[0], i
            for j in range(1, len(b) + 1):
                cur = min(D[j] + 1, D[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
                prev, D[j] = D[j], cur
        assert oracle.edit_distance(a, b) == D[-1]
    assert oracle.edit_distance(b"", b"ACG") == 3


def test_sink_tie_rule_agrees_with_exact_order(oracle):
    """The rule the HIP kernel uses instead of spoa's DFS order when several sinks share the best score
    (racon_amd/csrc/poa_kernel2.hpp, phase_sink_tie_rule) must pick the same sink as the exact rank order
    whenever it applies — on synthetic sets and on the reference's own data (tests/golden)."""
    import os
    from racon_amd.batch import WindowBatch
    from racon_amd.synth import simulate_windows
    sets = [(simulate_windows(60000, 500, 30, 10000, seed=20260921), (3, -5, -4)),
            (simulate_windows(40000, 500, 30, 10000, seed=5, with_quality=False), (3, -5, -4)),
            (simulate_windows(40000, 500, 30, 10000, seed=8), (1, -1, -1))]
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sets += [(WindowBatch.load(os.path.join(gold, "sam_fasta_w500.npz")), (5, -4, -8)),
             (WindowBatch.load(os.path.join(gold, "frag_kF_fastq_first200.npz")), (1, -1, -1))]
    oracle.tie_stats()
    events = ruled = 0
    for b, sc in sets:
        oracle.consensus(b, *sc, True, 0)
        e, r, a = oracle.tie_stats()
        assert r == a, f"rule disagrees with the exact order in {r - a} of {r} classified ties"
        events += e
        ruled += r
    assert events > 50 and ruled > 0.6 * events


def test_simd_baseline_variant_equals_scalar_oracle(oracle):
    """bench.py's CPU baseline is the oracle's AVX2 int16 variant; it must give the scalar oracle's bytes
    (including the rows that fall back to the scalar path: g = -100 does not fit int16)."""
    from helpers import edge_case_batch, synthetic_sets
    cases = [(edge_case_batch(), sc) for sc in [(3, -5, -4), (5, -4, -8), (1, -1, -1), (4, -6, -100)]]
    cases += [(b, sc) for _, b, sc in synthetic_sets()]
    for b, sc in cases:
        for trim in (True, False):
            r0 = oracle.consensus(b, *sc, trim, 0)
            r1 = oracle.consensus(b, *sc, trim, 0, simd=True)
            assert r0.consensus == r1.consensus and (r0.polished == r1.polished).all() and (r0.chimeric == r1.chimeric).all()
