"""The flat-array graph code the HIP kernel runs on one lane
(racon_amd/csrc/poa_core.hpp), compiled for the CPU with a scalar DP
(tests/emul/emul_main.cpp), against the oracle.  Catches logic errors in the
device data structures without a GPU.

The harness also restates, lane by lane, the Subgraph sweep the kernel runs instead of spoa's DFS (in-edge records ->
per-rank records -> 64-rank chunks with ring blocks, pending-rank visiting and chain runs) and compares its mask with the
DFS's for every partial layer (rc -5 on a difference), checks the in-edge records against the in-lists after every
layer (rc -4), and codes every alignment's cells as the banded DP does (one move byte per cell: first-argmax predecessors,
"diagonal / vertical reproduces the cell" flags), walks the codes alone and compares the path with the score traceback's
(rc -6)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import edge_case_batch, synthetic_sets

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "emul", "libemul.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "emul", "emul_main.cpp")])
    return C.CDLL(so)


def run_emul(lib, b, m, x, g, trim=True):
    cb = b.as_c()
    n = b.n_windows
    cap = int(b.bases.size) + 64
    off = np.zeros(n + 1, np.uint64)
    cons = np.zeros(cap, np.uint8)
    pol = np.zeros(n, np.uint8)
    rc = lib.rcn_emul_consensus(C.byref(cb), m, x, g, int(trim), off.ctypes.data_as(C.c_void_p),
                                cons.ctypes.data_as(C.c_void_p), C.c_uint64(cap), pol.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return [cons[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)], pol


def test_emul_edge_cases(emul, oracle):
    b = edge_case_batch()
    for sc in [(3, -5, -4), (1, -1, -1)]:
        ref = oracle.consensus(b, *sc, True, 2)
        got, pol = run_emul(emul, b, *sc)
        assert got == ref.consensus
        assert (pol == ref.polished).all()


@pytest.mark.parametrize("idx", range(7))
def test_emul_synthetic(emul, oracle, idx):
    name, b, sc = synthetic_sets()[idx]
    b = b.select(range(min(b.n_windows, 8)))
    ref = oracle.consensus(b, *sc, True, 4)
    got, pol = run_emul(emul, b, *sc)
    assert got == ref.consensus, name
    assert (pol == ref.polished).all()


def test_emul_serial_add_keeps_in_edge_records(emul, oracle, monkeypatch):
    """graph_add_alignment (add_node / add_edge) maintains the in-edge records as the per-position AddAlignment does:
    the emulator checks them against the in-lists after every layer (rc -4 on a difference)."""
    monkeypatch.setenv("RCN_EMUL_SERIAL_ADD", "1")
    name, b, sc = synthetic_sets()[0]
    b = b.select(range(min(b.n_windows, 6)))
    ref = oracle.consensus(b, *sc, True, 4)
    got, pol = run_emul(emul, b, *sc)
    assert got == ref.consensus


def test_emul_sink_tie_rule_on_fuzz_windows(emul, oracle):
    """The kernels' sink-tie rule (racon_amd/csrc/poa_k2_sinktie.hpp: phase_sink_tie_rule, the same rule in poa_small.hpp), restated in
    the emulator on the same arrays and compared with spoa's DFS order at EVERY tie (rc -8 on a difference), on the low-complexity fuzz
    windows where ties are the rule -- among them the window (tools/fuzz_sweep.py seed 5088, window 115, scores 1/-1/-1) on which the
    rule took a ring member OUTSIDE the Subgraph for the ring's backbone node: one window in 666 600 on the GPU, found by the sweep."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    from racon_amd.batch import WindowBatch
    from test_gpu_fuzz import random_window
    for seed, pick in ((5088, [115]), (5088, range(100, 260)), (7001, range(0, 160))):
        rng = np.random.default_rng(seed)
        wins = [random_window(rng, 5 * int(rng.integers(0, 50)) + (4 if rng.random() < 0.125 else int(rng.integers(0, 4)))) for _ in range(500)]
        b = WindowBatch.from_windows([wins[k] for k in pick])
        for sc in ((1, -1, -1), (3, -5, -4)):
            ref = oracle.consensus(b, *sc, True, 4)
            got, pol = run_emul(emul, b, *sc)
            assert got == ref.consensus, (seed, sc)
