"""-m gpu: the small-window kernel (racon_amd/csrc/poa_small.hpp: one wave per window, the graph in LDS) against the CPU
oracle and against poa_window_kernel2 -- BASELINE configs[3] (150-bp reads at 60x, -w 200: reference src/window.cpp:99-107,
the Subgraph branch nearly every layer takes) and every way a window can leave that kernel and come back through the
retry tier."""
import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from racon_amd.synth import config_windows, simulate_windows
from helpers import assert_same, edge_case_batch, q

pytestmark = pytest.mark.gpu

SHORT = dict(sub=0.003, ins=0.0005, dele=0.0005, phred_mean=30, phred_sd=0)


@pytest.fixture(scope="module")
def Engine():
    from racon_amd.engine import HipEngine
    return HipEngine


def _no_bug(st):
    assert st["small_bail_why"][8] == 0, st["small_bail_why"]          # internal inconsistencies: never


def test_cfg4_share_through_the_small_kernel(Engine, oracle):
    b = config_windows("cfg4", 0.04)                                    # 200 windows of ~140 layers
    ref = oracle.consensus(b, 3, -5, -4, True, 0)
    eng = Engine(3, -5, -4, True)
    got = eng.consensus(b)
    st = eng.stats()
    assert_same(got, ref, "cfg4 share")
    _no_bug(st)
    assert st["n_small"] + st["n_small_bailed"] == b.n_windows and st["n_small"] >= 0.9 * b.n_windows, st
    assert_same(eng.run(), ref, "resident batch again")
    assert_same(eng.consensus_refs(b), ref, "refs")


def test_small_kernel_equals_the_four_wave_kernel(Engine, monkeypatch):
    b = simulate_windows(30_000, 200, 60.0, 150, seed=41, **SHORT)
    eng = Engine(3, -5, -4, True)
    a = eng.consensus(b)
    assert eng.stats()["n_small"] > 0
    _no_bug(eng.stats())
    monkeypatch.setenv("RCN_NO_SMALL", "1")
    off = Engine(3, -5, -4, True)
    r = off.consensus(b)
    assert off.stats()["n_small"] == 0
    assert_same(a, r, "small kernel vs poa_window_kernel2")


@pytest.mark.parametrize("scores", [(3, -5, -4), (5, -4, -8), (1, -1, -1), (4, -6, -100)])
def test_scores_and_exact_consensus_order(Engine, oracle, monkeypatch, scores):
    b = simulate_windows(12_000, 200, 40.0, 150, seed=43, **SHORT)
    ref = oracle.consensus(b, *scores, True, 0)
    eng = Engine(*scores, True)
    assert_same(eng.consensus(b), ref, f"{scores}")
    _no_bug(eng.stats())
    monkeypatch.setenv("RCN_FORCE_EXACT", "1")                           # every window: spoa's own DFS order, then the bundle over it
    ex = Engine(*scores, True)
    assert_same(ex.consensus(b), ref, f"{scores}, exact order")
    _no_bug(ex.stats())


def test_tgs_trim_noisy_reads_and_the_way_back(Engine, oracle):
    """kTGS windows (coverage trim: the coverage atomics), reads noisy enough that graphs outgrow the LDS, get a fifth
    in-edge or a far predecessor: those windows come back flagged and poa_window_kernel2 polishes them."""
    b = simulate_windows(16_000, 200, 30.0, 2000, seed=45)              # ONT-like errors on 200-base windows
    ref = oracle.consensus(b, 3, -5, -4, True, 0)
    eng = Engine(3, -5, -4, True)
    got = eng.consensus(b)
    st = eng.stats()
    assert_same(got, ref, "noisy short windows")
    _no_bug(st)
    assert st["n_small_bailed"] > 0 and st["n_retried"] >= st["n_small_bailed"], st
    mild = simulate_windows(16_000, 200, 30.0, 2000, seed=46, sub=0.01, ins=0.005, dele=0.005)
    refm = oracle.consensus(mild, 3, -5, -4, True, 0)
    e2 = Engine(3, -5, -4, True)
    assert_same(e2.consensus(mild), refm, "mild TGS windows, trimmed")
    _no_bug(e2.stats())
    assert e2.stats()["n_small"] > 0
    notrim = Engine(3, -5, -4, False)
    assert_same(notrim.consensus(mild), oracle.consensus(mild, 3, -5, -4, False, 0), "mild TGS windows, no trim")


def test_no_quality_other_symbols_and_corner_windows(Engine, oracle):
    nq = simulate_windows(10_000, 200, 40.0, 150, seed=47, with_quality=False, **SHORT)
    eng = Engine(3, -5, -4, True)
    assert_same(eng.consensus(nq), oracle.consensus(nq, 3, -5, -4, True, 0), "no qualities")
    _no_bug(eng.stats())
    assert eng.stats()["n_small"] > 0
    # the hand-made corner windows (fewer than three sequences, begin ties, chimeric, lower case / N: other symbols leave the kernel)
    b = edge_case_batch()
    e2 = Engine(3, -5, -4, True)
    assert_same(e2.consensus(b), oracle.consensus(b, 3, -5, -4, True, 2), "edge cases")
    _no_bug(e2.stats())
    # a window with an N in one layer among ACGT windows (rings of up to five symbols stay in the kernel), and one whose
    # column 50 sees seven symbols (the ring outgrows the kernel's: that window comes back flagged)
    wins = []
    rng = np.random.default_rng(5)
    for k in range(70):
        bb = bytearray(rng.choice(list(b"ACGT"), 120).astype(np.uint8).tobytes())
        bb[60] = ord("A")
        bb = bytes(bb)
        lay = bb[10:110]
        seqs = [(bb, q(bb, 20), 0, 0)] + [(lay, q(lay, 25), 10, 109) for _ in range(4)] + [(bb, q(bb, 30), 0, 119)]
        if k == 13:
            withn = lay[:50] + b"N" + lay[51:]
            seqs[2] = (withn, q(withn, 25), 10, 109)
        if k == 31:
            for sym in b"CGTNRY":
                alt = lay[:50] + bytes([sym]) + lay[51:]
                seqs.append((alt, q(alt, 25), 10, 109))
        wins.append({"type": 0, "seqs": seqs})
    wb = WindowBatch.from_windows(wins)
    e3 = Engine(3, -5, -4, True)
    assert_same(e3.consensus(wb), oracle.consensus(wb, 3, -5, -4, True, 2), "windows with N / with a seven-symbol column")
    st = e3.stats()
    _no_bug(st)
    assert st["n_small_bailed"] == 1 and st["small_bail_why"][3] == 1 and st["n_small"] == wb.n_windows - 1, st


def test_long_queue_and_slot_counts(Engine, oracle):
    b = config_windows("cfg4", 0.02)
    big = b.select(list(range(b.n_windows)) * 50)                        # 5000 windows: more than the resident slots
    ref = oracle.consensus(b, 3, -5, -4, True, 0)
    eng = Engine(3, -5, -4, True)
    got = eng.consensus(big)
    _no_bug(eng.stats())
    for k in range(big.n_windows):
        assert got.consensus[k] == ref.consensus[k % b.n_windows], k
    few = Engine(3, -5, -4, True, max_slots=5)
    assert_same(few.consensus(b), ref, "five slots")


@pytest.mark.parametrize("seed,scores", [(11, (3, -5, -4)), (12, (5, -4, -8)), (13, (1, -1, -1)), (14, (2, -3, -2)), (15, (4, -6, -100)), (16, (3, -5, -4))])
def test_fuzz_small_windows(Engine, oracle, monkeypatch, seed, scores):
    """The tie-break stress windows of test_gpu_fuzz.py in the alphabets the kernel keeps (A/C/G/T and subsets: low-complexity
    backbones, co-optimal alignments, sink ties, weight ties in the heaviest bundle, zero qualities, duplicate begins, more
    than four in-edges per node -- the wide rows) through the small-window kernel: what it keeps equals the oracle, what it
    sends back (ties beyond the id rule, far predecessors ...) comes out of poa_window_kernel2 the same."""
    from test_gpu_fuzz import random_window
    rng = np.random.default_rng(2000 + seed)
    # styles 0-3: ACGT / AC / A / ACGTN (rings of up to five symbols stay in the kernel); one window in eight of style 4
    # (eight symbols: its rings can outgrow the kernel's, and the window then comes back through the retry tier)
    wins = [random_window(rng, 5 * int(rng.integers(0, 50)) + (4 if rng.random() < 0.125 else int(rng.integers(0, 4)))) for _ in range(500)]
    b = WindowBatch.from_windows(wins)
    for trim in (True, False):
        ref = oracle.consensus(b, *scores, trim, 0)
        eng = Engine(*scores, trim)
        assert_same(eng.consensus(b), ref, f"small fuzz seed {seed} {scores} trim={trim}")
        st = eng.stats()
        _no_bug(st)
        assert st["n_small"] > 0.5 * b.n_windows, st
    monkeypatch.setenv("RCN_FORCE_EXACT", "1")
    ex = Engine(*scores, True)
    assert_same(ex.consensus(b), oracle.consensus(b, *scores, True, 0), f"small fuzz seed {seed}, exact order")
    _no_bug(ex.stats())


@pytest.mark.parametrize("reads,ovl,scores", [("sample_reads.fastq.gz", "sample_overlaps.sam.gz", (3, -5, -4)),
                                              ("sample_reads.fastq.gz", "sample_overlaps.paf.gz", (5, -4, -8)),
                                              ("sample_reads.fasta.gz", "sample_overlaps.sam.gz", (1, -1, -1))])
def test_real_noisy_reads_at_w200(Engine, oracle, reads, ovl, scores):
    """REAL reads through the small-window kernel: the reference's own sample (test/data: ONT reads of a lambda phage assembly,
    ~10-15 % error) cut at `-w 200` -- layers of up to ~255 bases, graphs that outgrow the LDS, fifth-to-ninth in-edges, far
    predecessors: everything the synthetic short-read windows are too clean to produce (reference src/window.cpp:99-107 on the
    reference's data).  The product (Polisher::polish on the MI355X) prints the FASTA of host layer + oracle; the engine alone
    polishes the same windows with both kernels in play and no internal inconsistency."""
    from helpers import REFDATA as DATA
    from racon_amd import polisher as P
    P.build()

    def make():
        p = P.Polisher(DATA + reads, DATA + ovl, DATA + "sample_layout.fasta.gz", "kC", 200, 10, 0.3, True, *scores, num_threads=8)
        p.initialize()
        return p
    p = make()
    b = p.windows()
    assert b.n_windows > 200
    lens = np.diff(b.seq_off.astype(np.int64))
    assert 0 < int((lens > 255).sum()) * 8 < b.n_windows          # a few layers beyond the kernel's 255 bases: those windows are flagged at once
    ref = oracle.consensus(b, *scores, True, 0)
    ref_fasta = p.assemble(ref, True)
    p.close()
    eng = Engine(*scores, True)
    got = eng.consensus(b)
    st = eng.stats()
    assert_same(got, ref, f"sample reads at -w 200, {ovl}, {scores}")
    _no_bug(st)
    # the small kernel took the pass (a minority of windows outside its shape does not keep the others from it); what it sent back
    # -- graphs that outgrew the LDS, layers beyond 255 bases, far predecessors -- came back right through poa_window_kernel2
    # (real ONT reads at 30x grow a 200-base window's graph far beyond the kernel's LDS capacity: most windows come back)
    assert st["n_small"] + st["n_small_bailed"] > b.n_windows // 2 and st["n_small_bailed"] > 0 and st["n_retried"] >= st["n_small_bailed"], st
    p = make()
    assert p.polish(True) == ref_fasta
    p.close()


def test_bare_backbones_do_not_vote_for_the_small_kernel(Engine, oracle):
    """A window-range shard of a device-built job holds the other ranges' windows as bare backbones (fewer than three sequences: copied
    through, reference src/window.cpp:68-71).  Those have no shape to speak of: counted as "small", 1.75 M of them handed a batch of
    250 000 long-read windows to the small-window kernel, which flagged every real window (cfg5 whole in eight shards, round 5).  Only
    windows that will be polished count -- and a mostly-bare batch of SHORT-read windows still takes the kernel."""
    ont = simulate_windows(15_000, 500, 20.0, 3000, seed=61)                       # 30 windows with 500-base layers: outside the shape
    bare = [{"type": 1, "seqs": [(bytes([65 + (k % 3)]) * 500, None, 0, 0)]} for k in range(400)]
    wins = bare[:200] + [ont.window(k) for k in range(ont.n_windows)] + bare[200:]
    b = WindowBatch.from_windows(wins)
    ref = oracle.consensus(b, 3, -5, -4, True, 0)
    eng = Engine(3, -5, -4, True)
    assert_same(eng.consensus(b), ref, "bare backbones around long-read windows")
    st = eng.stats()
    assert st["n_small"] == 0 and st["n_small_bailed"] == 0, st                  # poa_window_kernel2 took the pass, nothing went through the retry tier
    short = simulate_windows(6_000, 200, 40.0, 150, seed=62, **SHORT)
    wins = bare[:300] + [short.window(k) for k in range(short.n_windows)]
    b2 = WindowBatch.from_windows(wins)
    e2 = Engine(3, -5, -4, True)
    assert_same(e2.consensus(b2), oracle.consensus(b2, 3, -5, -4, True, 0), "bare backbones around short-read windows")
    assert e2.stats()["n_small"] >= short.n_windows - e2.stats()["n_small_bailed"] > 0
