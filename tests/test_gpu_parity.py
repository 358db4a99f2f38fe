"""-m gpu: the HIP engine, called through the C ABI, against the CPU oracle.
Bar: byte-identical consensus strings and identical polished/chimeric flags
(integer scoring, deterministic tie-breaks)."""
import hashlib

import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_windows
from helpers import assert_same, edge_case_batch, edge_case_windows, synthetic_sets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from racon_amd.engine import HipEngine
    return HipEngine


def test_extension_is_the_hip_library(Engine):
    from racon_amd import engine
    lib = engine.load_library()
    assert lib.rcn_device_count() >= 1
    assert b"gfx950" in lib.rcn_version()


@pytest.mark.parametrize("scores", [(3, -5, -4), (5, -4, -8), (1, -1, -1)])
@pytest.mark.parametrize("trim", [True, False])
def test_edge_cases(Engine, oracle, scores, trim):
    b = edge_case_batch()
    ref = oracle.consensus(b, *scores, trim, 2)
    got = Engine(*scores, trim).consensus(b)
    assert_same(got, ref, f"edge {scores} trim={trim}")


@pytest.mark.parametrize("idx", range(7))
def test_synthetic_sets(Engine, oracle, idx):
    name, b, sc = synthetic_sets()[idx]
    ref = oracle.consensus(b, *sc, True, 0)
    eng = Engine(*sc, True)
    got = eng.consensus(b)
    assert_same(got, ref, name)
    assert eng.stats()["dp_cells"] > 0


def test_empty_batch_and_reuse(Engine, oracle):
    eng = Engine(3, -5, -4, True)
    empty = WindowBatch.from_windows([])
    r = eng.consensus(empty)
    assert r.consensus == []
    b = edge_case_batch()
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 1), "after empty")
    # same engine, second different batch, then the first again (buffers are reused)
    b2 = simulate_windows(5000, 500, 15, 3000, seed=9)
    assert_same(eng.consensus(b2), oracle.consensus(b2, 3, -5, -4, True, 0), "second batch")
    assert_same(eng.consensus(b), oracle.consensus(b, 3, -5, -4, True, 1), "first again")


def test_incremental_api_matches_batch_form(Engine, oracle):
    """addWindow / hasWindows / generateConsensus / reset (reference src/cuda/cudabatch.hpp:39-64)."""
    wins = edge_case_windows()
    eng = Engine(5, -4, -8, True)
    assert not eng.has_windows()
    for w in wins:
        assert eng.add_window(w)
    assert eng.has_windows()
    got = eng.generate_consensus()
    ref = oracle.consensus(WindowBatch.from_windows(wins), 5, -4, -8, True, 1)
    assert_same(got, ref, "incremental")
    eng.reset()
    assert not eng.has_windows()
    assert eng.add_window(wins[3])
    got = eng.generate_consensus()
    assert got.consensus == [ref.consensus[3]]


def test_capacity_retry_path(Engine, oracle):
    """Windows whose graphs outgrow the first-pass capacity estimate are re-run on the
    GPU with worst-case capacities (no CPU fallback).  Layers made of unrelated
    sequence force ~every base to become a new node."""
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    wins = []
    for k in range(3):
        bb = acgt[rng.integers(0, 4, 300)].tobytes()
        seqs = [(bb, b"!" * 300, 0, 0)]
        for _ in range(12):
            # low-complexity layers over disjoint 2-letter alphabets: almost no matches
            s = np.frombuffer(b"AC" if _ % 2 else b"GT", np.uint8)[rng.integers(0, 2, 290)].tobytes()
            seqs.append((s, None, 0, 299))
        wins.append({"type": 1, "seqs": seqs})
    wins += edge_case_windows()[:4]
    b = WindowBatch.from_windows(wins)
    ref = oracle.consensus(b, 3, -5, -4, True, 0)
    eng = Engine(3, -5, -4, True)
    got = eng.consensus(b)
    assert_same(got, ref, "retry")


def test_slot_count_and_arena_do_not_change_results(Engine, oracle):
    name, b, sc = synthetic_sets()[0]
    ref = oracle.consensus(b, *sc, True, 0)
    for slots in (1, 3, 64):
        assert_same(Engine(*sc, True, max_slots=slots).consensus(b), ref, f"slots={slots}")
    assert_same(Engine(*sc, True, arena_bytes=64 << 20).consensus(b), ref, "small arena")


def test_work_groups_per_cu_do_not_change_results(Engine, oracle, monkeypatch):
    """The engine runs a batch that is resident all at once with six work-groups per CU instead of eight
    (engine.hip: wg_per_cu); the choice, and any forced value, must not show in the output."""
    b = simulate_windows(300_000, 500, 30.0, 10000, seed=77)             # 600 windows: all resident, the deepest one bounds the launch
    monkeypatch.setenv("RCN_SPLIT", "0")                                 # (the split launch has its own tests: test_gpu_product_path.py)
    eng = Engine(3, -5, -4, True)
    base = eng.consensus(b)
    assert eng.stats()["wg_per_cu"] == 6
    perm = np.random.default_rng(5).permutation(b.n_windows)[:120]
    assert_same(Engine(3, -5, -4, True).consensus(b.select(perm)), oracle.consensus(b.select(perm), 3, -5, -4, True, 0), "sample")
    for n in (8, 7, 5, 4):
        monkeypatch.setenv("RCN_WG_PER_CU", str(n))
        e2 = Engine(3, -5, -4, True)
        r = e2.consensus(b)
        assert e2.stats()["wg_per_cu"] == n
        assert r.consensus == base.consensus and list(r.polished) == list(base.polished), n
    monkeypatch.delenv("RCN_WG_PER_CU")
    # a long queue (more windows than slots, none much deeper than its share) keeps eight
    big = b.select(list(range(b.n_windows)) * 8)
    e3 = Engine(3, -5, -4, True)
    r = e3.consensus(big)
    assert e3.stats()["wg_per_cu"] == 8
    assert r.consensus[:b.n_windows] == base.consensus and r.consensus[-b.n_windows:] == base.consensus


def test_full_size_properties(Engine, oracle):
    """BASELINE.json configs[1] at full size (2000 windows): size-independent
    properties — idempotence, independence from batch composition / order, and a
    checksum-of-checksums against the oracle on a seeded sample."""
    b = simulate_windows(1_000_000, 500, 30.0, 10000, seed=20260921)
    assert b.n_windows == 2000
    eng = Engine(3, -5, -4, True)
    r1 = eng.consensus(b)
    r2 = eng.run()
    assert r1.consensus == r2.consensus                                    # idempotent on resident inputs
    perm = np.random.default_rng(0).permutation(b.n_windows)[:300]
    sub = b.select(perm)
    r3 = Engine(3, -5, -4, True).consensus(sub)                            # different batch composition + order
    assert [r1.consensus[i] for i in perm] == r3.consensus
    ref = oracle.consensus(sub, 3, -5, -4, True, 0)
    assert_same(r3, ref, "cfg2 sample")
    h = hashlib.sha256()
    for c in r1.consensus:
        h.update(hashlib.sha256(c).digest())
    # every window polished, consensus length close to the window length
    assert all(r1.polished)
    lens = np.array([len(c) for c in r1.consensus])
    assert 450 < lens[:-1].mean() < 520
    print("cfg2 checksum-of-checksums", h.hexdigest())


def test_int32_fallback_kernel(Engine, oracle):
    """Scores whose Z-domain bound does not fit int16 (|g| * V > 31000) leave poa_window_kernel2 after the
    validity check and are re-run by the int32 kernel with worst-case capacities (no CPU fallback anywhere)."""
    b = simulate_windows(6000, 500, 20, 3000, seed=21)
    sc = (4, -6, -100)
    eng = Engine(*sc, True)
    got = eng.consensus(b)
    assert_same(got, oracle.consensus(b, *sc, True, 0), "int32 fallback")
    deep = int((np.diff(b.win_seq_off) >= 3).sum())
    assert eng.stats()["n_retried"] == deep


def test_one_wave_and_pipeline_dp_agree(Engine, oracle, monkeypatch):
    """RCN_HEAVY_PCT=0 sends every window through the 4-wave mailbox pipeline DP instead of the one-wave DP."""
    b = simulate_windows(10000, 500, 25, 4000, seed=33)
    ref = oracle.consensus(b, 3, -5, -4, True, 0)
    monkeypatch.setenv("RCN_HEAVY_PCT", "0.0")
    assert_same(Engine(3, -5, -4, True).consensus(b), ref, "pipeline DP")
    monkeypatch.setenv("RCN_HEAVY_PCT", "1.0")
    assert_same(Engine(3, -5, -4, True).consensus(b), ref, "one-wave DP")
