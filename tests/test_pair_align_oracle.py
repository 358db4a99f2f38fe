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


def test_device_cell_form_equals_the_textbook_recurrence(tmp_path):
    """racon_amd/csrc/pair_cell.hpp -- the cell of the device aligner as it is issued on gfx950 (three-input bit operations, the
    horizontal plus-word complemented, carries handed over in bit 31) -- compiled for the CPU: every word of every column equals the
    64-bit textbook form (state, stored deltas, carries), and the column scores equal a plain edit-distance DP, for 2 / 3 / 8 symbol
    planes and symbols the query lacks (tests/emul/pair_cell_main.cpp)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pair_cell_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(here, "emul", "pair_cell_main.cpp")])
    out = subprocess.run([exe, "150"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "pair_cell ok" in out.stdout
