"""-m gpu: the product's self-check (rcn_engine_verify, include/racon_hip.h; RACON_HIP_VERIFY of the host layer) and real reads at
more shapes.

The consensus kernels replace plain work of spoa by proved shortcuts -- an exact band with a certificate, move codes instead of the
score matrix, a rule instead of spoa's DFS order at tied sinks and in the consensus, a rank-interval Subgraph, a one-wave kernel for
small windows -- and twice a shortcut shipped wrong (rounds 4 and 5: found by fuzzing, one window in 666 600).  The self-check polishes
a sample of every batch a second time ON THE GPU with all of them switched off and compares: no oracle in the product, and a
difference is fatal.  Here: it passes where the kernels are right, and it catches a PLANTED wrong rule (RCN_PLANT_FAULT=1: the
sink-tie rule picks the last key instead of the first) -- exactly the windows the oracle says are wrong."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import REFDATA as DATA, assert_same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "racon_amd", "host", "racon_hip")


@pytest.fixture(scope="module")
def Engine(hip_lib):
    from racon_amd.engine import HipEngine
    return HipEngine


def test_self_check_passes_on_every_kind_of_window(Engine, oracle):
    from helpers import edge_case_batch, synthetic_sets
    from racon_amd.batch import WindowBatch
    for name, batch, scores in synthetic_sets() + [("edge cases", edge_case_batch(), (3, -5, -4))]:
        eng = Engine(*scores, True)
        got = eng.consensus(batch)
        st = eng.stats()
        rep = eng.verify(1.0)
        polishable = int((np.diff(batch.win_seq_off.astype(np.int64)) >= 3).sum())
        assert rep["n_checked"] == polishable and rep["n_differ"] == 0 and rep["first_window"] == 0xffffffff, (name, rep)
        # the check neither changes the run's results nor its statistics
        assert_same(eng.result(), got, name)
        assert eng.stats()["dp_cells"] == st["dp_cells"] and eng.stats()["kernel_ms"] == st["kernel_ms"]
        assert_same(got, oracle.consensus(batch, *scores, True, 0), name)
        # a fraction: the same windows every time, about that share
        r1, r2 = eng.verify(0.25), eng.verify(0.25)
        assert r1["n_checked"] == r2["n_checked"] and 1 <= r1["n_checked"] <= max(1, polishable) and r1["n_differ"] == 0


def test_self_check_catches_a_planted_wrong_rule(Engine, oracle, monkeypatch):
    from racon_amd.synth import simulate_windows
    batch = simulate_windows(150_000, 500, 30.0, 10_000, seed=77)          # 300 ONT-like windows: dozens of sink ties
    ref = oracle.consensus(batch, 3, -5, -4, True, 0)
    monkeypatch.setenv("RCN_PLANT_FAULT", "1")
    bad = Engine(3, -5, -4, True)                                         # (switches are read when the engine is created)
    monkeypatch.delenv("RCN_PLANT_FAULT")
    got = bad.consensus(batch)
    wrong = [i for i in range(batch.n_windows) if got.consensus[i] != ref.consensus[i] or got.polished[i] != ref.polished[i]]
    assert wrong, "the planted rule changed nothing: the test has no teeth on this batch"
    rep = bad.verify(1.0)
    assert rep["n_differ"] == len(wrong) and rep["first_window"] == wrong[0], (rep, wrong[:5])
    # a sample catches its share: the sampled wrong windows, no others
    rep2 = bad.verify(0.5)
    assert 0 <= rep2["n_differ"] <= len(wrong) and (rep2["n_differ"] == 0 or rep2["first_window"] in wrong)
    # the same batch on a sound engine passes
    good = Engine(3, -5, -4, True)
    assert_same(good.consensus(batch), ref, "sound engine")
    assert good.verify(1.0)["n_differ"] == 0


def test_racon_hip_verify_option(tmp_path):
    """RACON_HIP_VERIFY=<fraction> through the binary: same FASTA, exit 0; with the planted rule: a fatal error that names a window --
    on host-built windows (chunks) and on device-built ones."""
    sys.path.insert(0, ROOT)
    import bench
    files = bench.product_files(300_000, 30.0, 20260931, 8)
    cmd = [EXE, "-t", "8", files["reads"], files["sam"], files["targets"]]
    base = {k: v for k, v in os.environ.items() if not k.startswith("RCN_") and k != "RACON_HIP_VERIFY"}
    plain = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=base)
    assert plain.returncode == 0
    for mode in ("0", "auto"):
        env = dict(base, RACON_HIP_VERIFY="0.5", RACON_HIP_DEVICE_WINDOWS=mode, RACON_HIP_TIMING="1")
        ok = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert ok.returncode == 0 and ok.stdout == plain.stdout, ok.stderr[-400:]
        assert b"self-check:" in ok.stderr and b"none differs" in ok.stderr
        env = dict(base, RACON_HIP_VERIFY="1", RACON_HIP_DEVICE_WINDOWS=mode, RCN_EXPERIMENT="1", RCN_PLANT_FAULT="1")
        caught = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert caught.returncode == 1 and b"self-check failed" in caught.stderr and b"first: window" in caught.stderr, caught.stderr[-400:]
        assert caught.stdout == b""


SCORES = [(3, -5, -4), (5, -4, -8), (1, -1, -1), (2, -3, -2), (4, -6, -100)]


@pytest.mark.parametrize("w", [100, 150, 300, 700, 1500])
@pytest.mark.parametrize("typ", ["kC", "kF"])
def test_real_reads_at_more_shapes(oracle, monkeypatch, typ, w):
    """The reference's own sample (test/data: ONT reads of a lambda phage assembly; contig mode from the SAM overlaps, fragment
    correction from the all-versus-all PAF) at window lengths the synthetic sets do not have, x five score sets: the PRODUCT
    (Polisher::polish on the MI355X; host-built chunks, device-built windows, device aligner in turn) prints the FASTA of host layer +
    oracle (reference src/window.cpp:88-107, test/racon_test.cpp:86-295 parameterise -w 500 / 1000 only)."""
    from racon_amd import polisher as P
    P.build()
    reads = DATA + "sample_reads.fastq.gz"
    ovl, targets = (DATA + "sample_overlaps.sam.gz", DATA + "sample_layout.fasta.gz") if typ == "kC" else (DATA + "sample_ava_overlaps.paf.gz", reads)
    p0 = P.Polisher(reads, ovl, targets, typ, w, 10, 0.3, True, *SCORES[0], num_threads=16)
    p0.initialize()                                 # (host-built windows: what the oracle polishes; they do not depend on the scores)
    b = p0.windows()
    dev = "2" if typ == "kC" else "3"
    for k, scores in enumerate(SCORES):
        ref = oracle.consensus(b, *scores, True, 0)
        if k == 0:
            holder = p0
        else:
            # (stitching only needs the backbones, names and coverages: a Polisher that leaves alignment and construction to the device
            #  -- and, told so, to polish() -- has them without the host's pairwise alignment of every overlap)
            monkeypatch.setenv("RACON_HIP_DEVICE_WINDOWS", dev); monkeypatch.setenv("RACON_HIP_BUILD_IN_POLISH", "1")
            holder = P.Polisher(reads, ovl, targets, typ, w, 10, 0.3, True, *scores, num_threads=16)
            holder.initialize()
            monkeypatch.delenv("RACON_HIP_DEVICE_WINDOWS"); monkeypatch.delenv("RACON_HIP_BUILD_IN_POLISH")
            assert holder.num_windows() == b.n_windows
        ref_fasta = holder.assemble(ref, typ == "kC")
        holder.close()
        # the product, construction path in turn: host-built chunks / CIGAR walk (kC) or pairwise alignment (kF) + construction in HBM
        mode = ("0", dev, "auto")[k % 3]
        monkeypatch.setenv("RACON_HIP_DEVICE_WINDOWS", mode)
        monkeypatch.setenv("RACON_HIP_VERIFY", "0.1")
        q = P.Polisher(reads, ovl, targets, typ, w, 10, 0.3, True, *scores, num_threads=16)
        q.initialize()
        fasta = q.polish(typ == "kC")
        q.close()
        monkeypatch.delenv("RACON_HIP_DEVICE_WINDOWS"); monkeypatch.delenv("RACON_HIP_VERIFY")
        assert fasta == ref_fasta, (typ, w, scores, mode)
