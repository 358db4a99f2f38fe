"""tests/golden/: packed windows of the reference's own test data (built by tools/make_golden.py
in the container that has /root/reference) with the oracle's per-window consensus and the golden
numbers of reference test/racon_test.cpp.  The fixtures travel to the GPU box, /root/reference does not.

not gpu: the oracle reproduces the manifest (per-window md5s) and, through them, the reference's
         golden edit distances (racon_test.cpp:151,174,197,220).
gpu    : the HIP engine (C ABI) is byte-identical on every fixture window and hits the same goldens.
"""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from racon_amd.batch import WindowBatch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
CASES = sorted(MANIFEST["cases"])


def revcomp(s: bytes) -> bytes:
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


def _check(name, res, edit_distance):
    c = MANIFEST["cases"][name]
    assert len(res.consensus) == c["n_windows"]
    bad = [i for i, s in enumerate(res.consensus) if hashlib.md5(s).hexdigest() != c["consensus_md5"][i]]
    assert not bad, f"{name}: {len(bad)} windows differ from the golden consensus, first {bad[:5]}"
    assert [int(v) for v in res.polished] == c["polished"]
    assert [int(v) for v in res.chimeric] == c["chimeric"]
    if "golden_edit_distance" in c:
        # contig polishing of one target: polished sequence = concatenation of the window consensi
        # (reference src/polisher.cpp:510-531); the golden is the edit distance of its reverse
        # complement to sample_reference (reference test/racon_test.cpp:98-106)
        polished = b"".join(res.consensus)
        assert hashlib.md5(polished).hexdigest() == c["polished_md5"]
        ref = gzip.open(os.path.join(GOLD, "reference_contig.txt.gz")).read()
        assert edit_distance(revcomp(polished), ref) == c["golden_edit_distance"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_fixture(oracle, name):
    c = MANIFEST["cases"][name]
    b = WindowBatch.load(os.path.join(GOLD, name + ".npz"))
    res = oracle.consensus(b, *c["scores"], c["trim"], 0)
    _check(name, res, oracle.edit_distance)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_fixture(oracle, name):
    from racon_amd.engine import HipEngine
    c = MANIFEST["cases"][name]
    b = WindowBatch.load(os.path.join(GOLD, name + ".npz"))
    eng = HipEngine(*c["scores"], c["trim"])
    res = eng.consensus(b)
    _check(name, res, oracle.edit_distance)      # the oracle only serves as the edit-distance helper here
    assert eng.stats()["dp_cells"] > 0


@pytest.mark.gpu
def test_hip_fixture_through_incremental_abi():
    """addWindow/generateConsensus form of the ABI (reference src/cuda/cudabatch.hpp:39-64) on real windows."""
    from racon_amd.engine import HipEngine
    name = "sam_fastq_w500"
    c = MANIFEST["cases"][name]
    b = WindowBatch.load(os.path.join(GOLD, name + ".npz")).select(range(24))
    eng = HipEngine(*c["scores"], c["trim"])
    for w in range(b.n_windows):
        assert eng.add_window(b.window(w))
    assert eng.has_windows()
    res = eng.generate_consensus()
    eng.reset()
    assert not eng.has_windows()
    assert [hashlib.md5(s).hexdigest() for s in res.consensus] == c["consensus_md5"][:24]
