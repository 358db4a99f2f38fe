"""not gpu: the alignment oracle (oracle/nw_oracle.py = SURVEY Appendix B's validated rule) against the host layer's
aligner (racon_amd/host/nw_path.cpp, which reproduces the PAF / MHAP goldens end to end): same CIGAR, byte for byte, on
pairs below and above the 1 MiB traceback-state threshold where the rule switches to Hirschberg."""
import numpy as np
import pytest

from pairgen import mutate, random_seq


@pytest.fixture(scope="module")
def P():
    from racon_amd import polisher
    polisher.build()
    return polisher


@pytest.mark.parametrize("seed,n,rate", [(1, 40, 0.2), (2, 700, 0.1), (3, 1800, 0.15), (4, 2300, 0.12), (5, 5200, 0.1)])
def test_oracle_matches_host_aligner(P, seed, n, rate):
    from oracle import nw_oracle
    rng = np.random.default_rng(9000 + seed)
    t = random_seq(rng, n)
    q = mutate(rng, t, rate)
    c, d = nw_oracle.cigar(q, t)
    assert c == P.align_cigar(q, t).encode()
    assert d == P.edit_distance(q, t)


def test_oracle_edge_cases(P):
    from oracle import nw_oracle
    for q, t in [(b"A", b"A"), (b"A", b"C"), (b"ACGT", b"A"), (b"A", b"ACGT"), (b"ACGTACGT", b"ACGTACGT"), (b"AAAA", b"TTTT"),
                 (b"ACGTNNACGT", b"ACGTACGT"), (b"GATTACA" * 30, b"GATACA" * 30)]:
        assert nw_oracle.cigar(q, t)[0] == P.align_cigar(q, t).encode(), (q, t)
    assert nw_oracle.reverse_complement(b"AACGTN") == b"NACGTT"
