"""-m gpu: the paths the PRODUCT takes through the engine (round 3) against the oracle.

* rcn_engine_polish_refs -- the batch as borrowed per-sequence pointers, what a racon::Window holds (reference
  src/window.hpp:71-73) -- in its plain (< 64 windows), streamed and queued (more windows than slots) forms;
* the split launch (engine.hip: split_plan): deepest windows on CUs of their own, forced on, forced off, other CU counts;
* rcn_engine_reserve ahead of the first batch, too small and too large;
* rcn_engine_export_batch after a streamed batch (device layout is deepest first, the copy is in caller order);
* the code waves of a window that has a CU to itself (poa_band.hpp), in the deep launch and with every window alone on a CU;
* Polisher::polish on files: engines created by initialize()'s warm-up thread, chunks in deepest-first order over two
  engines, the Logger-bracketed interval reported through the C ABI -- same FASTA as host layer + oracle.
"""
import os

import numpy as np
import pytest

from racon_amd.synth import simulate_window_files, simulate_windows
from helpers import assert_same, edge_case_batch, synthetic_sets

pytestmark = pytest.mark.gpu

FIELDS = ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals")


@pytest.fixture(scope="module")
def Engine():
    from racon_amd.engine import HipEngine
    return HipEngine


@pytest.fixture(scope="module")
def mid():
    """600 ONT-like windows, all resident at once, the deepest one bounds the launch."""
    return simulate_windows(300_000, 500, 30.0, 10000, seed=77)


@pytest.fixture(scope="module")
def mid_ref(mid, oracle):
    return oracle.consensus(mid, 3, -5, -4, True, 0, simd=True)


def test_refs_form_small_batches(Engine, oracle):
    for scores in ((3, -5, -4), (5, -4, -8)):
        b = edge_case_batch()                          # < 64 windows: the plain upload path behind the pointer form
        assert_same(Engine(*scores, True).consensus_refs(b), oracle.consensus(b, *scores, True, 2), "edge windows as refs")
    name, b, sc = synthetic_sets()[3]                  # layers without quality: NULL quality pointers
    layers = np.ones(b.n_seqs, bool); layers[b.win_seq_off[:-1]] = False
    assert int(b.seq_has_qual[layers].max()) == 0 and layers.sum() > 100
    assert_same(Engine(*sc, True).consensus_refs(b), oracle.consensus(b, *sc, True, 0), name + " as refs")


def test_refs_form_streamed_and_queued(Engine, mid, mid_ref):
    from racon_amd.engine import REFS_QUEUED
    eng = Engine(3, -5, -4, True)
    assert_same(eng.consensus_refs(mid), mid_ref, "600 windows as refs")
    assert_same(eng.consensus(mid), mid_ref, "the same engine, batch form")
    r = eng.consensus_refs(mid, REFS_QUEUED)
    assert_same(r, mid_ref, "queued")
    st = eng.stats()
    assert st["wg_per_cu"] == 8 and st["split_deep"] == 0          # part of a longer queue: full residency, no split
    # more windows than resident slots: the launches are persistent over their queues
    small = Engine(3, -5, -4, True, max_slots=48)
    assert_same(small.consensus_refs(mid), mid_ref, "48 slots")
    assert_same(small.consensus(mid), mid_ref, "48 slots, batch form")


@pytest.mark.parametrize("cus,per_cu", [("", ""), ("32", "2"), ("128", "1")])
def test_split_launch_on_and_off(Engine, mid, mid_ref, monkeypatch, cus, per_cu):
    if cus:
        monkeypatch.setenv("RCN_SPLIT_CUS", cus)
        monkeypatch.setenv("RCN_SPLIT_DEEP_PER_CU", per_cu)
    monkeypatch.setenv("RCN_SPLIT", "1")
    on = Engine(3, -5, -4, True)
    r = on.consensus(mid)                               # streamed: piece 0 is the deep launch
    st = on.stats()
    assert st["split_deep"] > 0 and st["n_launches"] == 2 and st["split_cus"] == (int(cus) if cus else 32)
    assert st["launch_ms"][0] > 0 and st["launch_ms"][1] > 0 and st["kernel_ms"] >= max(st["launch_ms"]) * 0.999
    assert_same(r, mid_ref, "split, streamed")
    assert_same(on.run(), mid_ref, "split, resident batch (deepest-first layout)")
    on.upload(mid)
    assert_same(on.run(), mid_ref, "split, resident batch (caller layout)")
    assert on.stats()["split_deep"] == st["split_deep"]
    assert_same(on.consensus_refs(mid), mid_ref, "split, refs")
    monkeypatch.setenv("RCN_SPLIT", "0")
    off = Engine(3, -5, -4, True)
    assert_same(off.consensus(mid), mid_ref, "no split")
    assert off.stats()["split_deep"] == 0 and off.stats()["n_launches"] == 2        # (the two streamed pieces)
    off.upload(mid)
    assert_same(off.run(), mid_ref, "no split, resident")
    assert off.stats()["split_deep"] == 0 and off.stats()["n_launches"] == 1


def test_split_launch_rule(Engine, mid, mid_ref):
    """Left to itself the engine splits a batch that is resident at once and ruled by its deepest window, and does not
    split a long queue."""
    eng = Engine(3, -5, -4, True)
    eng.upload(mid)
    assert_same(eng.run(), mid_ref, "rule")
    assert eng.stats()["split_deep"] > 0
    big = mid.select(list(range(mid.n_windows)) * 8)
    r = eng.consensus(big)
    assert eng.stats()["split_deep"] == 0 and eng.stats()["wg_per_cu"] == 8
    assert r.consensus[:mid.n_windows] == mid_ref.consensus and r.consensus[-mid.n_windows:] == mid_ref.consensus


def test_reserve_ahead_of_the_first_batch(Engine, mid, mid_ref):
    small = Engine(3, -5, -4, True)
    small.reserve(64, 64 * 10, 64 * 10 * 500, 500)                    # far too small: the batch grows everything
    assert_same(small.consensus_refs(mid), mid_ref, "after a small reservation")
    large = Engine(3, -5, -4, True)
    large.reserve(2048, 2048 * 41, 2048 * 41 * 500, 500, 700, 45 * 500)
    assert_same(large.consensus_refs(mid), mid_ref, "after a large reservation")
    assert_same(large.consensus(edge_case_batch()), Engine(3, -5, -4, True).consensus(edge_case_batch()), "other shapes afterwards")
    large.reserve(0, 0, 0, 500)                                       # warm-up only


def test_export_batch_is_in_caller_order_after_a_streamed_batch(Engine, mid):
    eng = Engine(3, -5, -4, True)
    eng.consensus(mid)                                                 # resident deepest first
    got = eng.export_batch()
    mid.as_c()
    for f in FIELDS:
        assert np.array_equal(getattr(got, f), getattr(mid, f)), f
    eng.upload(mid)                                                    # resident in caller order
    got = eng.export_batch()
    for f in FIELDS:
        assert np.array_equal(getattr(got, f), getattr(mid, f)), f


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("product"))
    return simulate_window_files(d, 700_000, 30.0, 10000, seed=20260921, piece=300_000, workers=3)


def test_polish_interval_chunks_and_warmup(files, oracle, monkeypatch):
    from racon_amd import polisher as P
    P.build()

    def make():
        return P.Polisher(files["reads"], files["sam"], files["targets"], "kC", 500, 10.0, 0.3, True, 3, -5, -4, num_threads=8)
    p = make()
    p.initialize()
    b = p.windows()
    assert b.n_windows == 1400
    ref = p.assemble(oracle.consensus(b, 3, -5, -4, True, 0, simd=True), True)
    p.close()
    p = make(); p.initialize()
    assert p.polish(True) == ref                                       # one chunk, engines from the warm-up thread
    assert p.polish_plan() == (1, 1)
    t_warm = p.polish_seconds()
    assert 0.0 < t_warm < 5.0
    p.close()
    for chunk in ("100", "333"):                                       # 14 / 5 chunks in deepest-first order over two engines
        monkeypatch.setenv("RACON_HIP_CHUNK_WINDOWS", chunk)
        p = make(); p.initialize()
        assert p.polish(True) == ref, chunk
        assert p.polish_plan() == ((1400 + int(chunk) - 1) // int(chunk), 2), chunk      # RACON_HIP_CHUNK_WINDOWS is exact: no byte floor
        p.close()
    monkeypatch.delenv("RACON_HIP_CHUNK_WINDOWS")
    monkeypatch.setenv("RACON_HIP_NO_WARMUP", "1")                      # engines created inside polish() (the round-2 product)
    p = make(); p.initialize()
    assert p.polish(True) == ref
    assert p.polish_seconds() > 0.0
    p.close()


def test_matrix_rows_smaller_than_the_graph(Engine, mid, mid_ref, monkeypatch):
    """A slot's DP matrix is sized for fewer rows than its graph arrays hold nodes (engine.hip: first_pass_caps); an
    alignment that needs more flags the window and the retry pass (worst-case capacities) takes it.  RCN_HROWS_DIV=400
    leaves room for the backbone and ~160 more rows: most ONT-like windows outgrow that after a few layers."""
    monkeypatch.setenv("RCN_HROWS_DIV", "400")
    eng = Engine(3, -5, -4, True)
    sub = mid.select(range(96))
    got = eng.consensus(sub)
    assert eng.stats()["n_retried"] > 48
    assert got.consensus == mid_ref.consensus[:96] and list(got.polished) == list(mid_ref.polished[:96])


def test_noisy_reads_raise_the_capacity_estimates(Engine, oracle):
    """Reads at 25 % error add a node per ~4 layer bases -- beyond what the first-pass estimates leave room for (a node per
    four, a matrix row per six): many windows go to the retry pass, the results do not change, and the engine gives the NEXT
    batch more room (engine.hip: caps_level)."""
    b = simulate_windows(100_000, 500, 30.0, 10000, seed=78, sub=0.08, ins=0.08, dele=0.09)
    ref = oracle.consensus(b, 3, -5, -4, True, 0, simd=True)
    eng = Engine(3, -5, -4, True)
    assert_same(eng.consensus(b), ref, "25 % error, first batch")
    first = eng.stats()["n_retried"]
    assert_same(eng.consensus_refs(b), ref, "25 % error, second batch")
    second = eng.stats()["n_retried"]
    assert_same(eng.consensus(b), ref, "25 % error, third batch")
    third = eng.stats()["n_retried"]
    print("windows sent to the retry pass: %d, %d, %d of %d" % (first, second, third, b.n_windows))
    assert third <= second <= first, (first, second, third)
    if first > b.n_windows // 50:                       # the estimates were too small: they must have grown
        assert third < first, (first, second, third)


def test_code_waves_of_the_deep_launch(Engine, mid, mid_ref, oracle, monkeypatch):
    """A window with a CU (and its LDS) to itself runs the banded DP with code waves (poa_band.hpp: waves 1-3 assemble the
    move codes of the chain / fast rows from the large LDS ring): same bytes as without them, in the deep launch of the
    split and with EVERY window alone on a CU -- ONT-like windows (window shifts, predecessors written under older
    offsets), noisy backbones, forced failures of the certificate, forced sink-tie levels."""
    monkeypatch.setenv("RCN_SPLIT", "1")
    eng = Engine(3, -5, -4, True)
    assert_same(eng.consensus(mid), mid_ref, "deep launch with code waves")
    st = eng.stats()
    assert st["split_deep"] > 0 and st["n_code_wave"] > 20 * st["split_deep"]      # ~30 banded alignments per deep window
    monkeypatch.setenv("RCN_NO_CODE_WAVE", "1")
    off = Engine(3, -5, -4, True)
    assert_same(off.consensus(mid), mid_ref, "deep launch without")
    assert off.stats()["n_code_wave"] == 0 and off.stats()["n_banded"] == st["n_banded"]
    monkeypatch.delenv("RCN_NO_CODE_WAVE")
    # every window alone on a CU: every banded alignment goes through the code waves
    monkeypatch.setenv("RCN_SPLIT", "0")
    monkeypatch.setenv("RCN_WG_PER_CU", "1")
    one = Engine(3, -5, -4, True)
    assert_same(one.consensus(mid), mid_ref, "one work-group per CU")
    s1 = one.stats()
    assert s1["wg_per_cu"] == 1 and s1["n_code_wave"] == s1["n_banded"] == st["n_banded"]
    for name, b, (m, x, g) in synthetic_sets():
        ref = oracle.consensus(b, m, x, g, True, 0, simd=True)
        e = Engine(m, x, g, True)
        assert_same(e.consensus(b), ref, name)
        if name.startswith("ont_w500") or name == "noisy_backbone":
            assert e.stats()["n_code_wave"] == e.stats()["n_banded"] > 0, name
    for var, val in (("RCN_FORCE_BAND_FAIL", "1"), ("RCN_FORCE_TIE", "2"), ("RCN_FORCE_TIE", "3"), ("RCN_FORCE_SLOW_TB", "1")):
        monkeypatch.setenv(var, val)
        sub = mid.select(list(range(0, mid.n_windows, 7)))
        r = Engine(3, -5, -4, True).consensus(sub)
        for k, i in enumerate(range(0, mid.n_windows, 7)):
            assert r.consensus[k] == mid_ref.consensus[i], (var, val, i)
        monkeypatch.delenv(var)
