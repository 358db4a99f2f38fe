"""-m gpu, BASELINE.json's cfg2 at full size (1 Mbp contig, 30x, w=500: 2000 windows, 24.5 G DP cells), through
properties that do not need the oracle at that size:
  * determinism / idempotence: two launches over the resident batch give identical bytes;
  * independence (what sharding over GPUs relies on, reference src/polisher.cpp:496-503): polishing a shard alone
    gives exactly the slice of the full result, for an uneven 3-way split;
  * a checksum of checksums over all windows equals the one of the sharded runs;
  * spot check: a seeded sample of windows against the oracle.
"""
import hashlib

import numpy as np
import pytest

from racon_amd.synth import config_windows

pytestmark = pytest.mark.gpu


def _digest(cons):
    h = hashlib.sha256()
    for c in cons:
        h.update(hashlib.md5(c).digest())
    return h.hexdigest()


def test_cfg2_full_size_properties(oracle):
    from racon_amd.engine import HipEngine
    b = config_windows("cfg2")
    assert b.n_windows == 2000
    eng = HipEngine(3, -5, -4, True)
    eng.upload(b)
    r1 = eng.run()
    r2 = eng.run()
    assert r1.consensus == r2.consensus and (r1.polished == r2.polished).all()
    st = eng.stats()
    assert st["n_retried"] == 0 and st["dp_cells"] > 2.0e10
    # shards: each rank's result is the slice of the full result
    parts = []
    for rank in range(3):
        sub, idx = b.shard(rank, 3)
        rs = HipEngine(3, -5, -4, True).consensus(sub)
        assert rs.consensus == [r1.consensus[int(i)] for i in idx]
        parts += rs.consensus
    assert _digest(parts) == _digest(r1.consensus)
    # consensus lengths stay near the window length (TGS trimming may shorten the ends)
    lens = np.array([len(c) for c in r1.consensus])
    assert 400 < np.median(lens) < 520
    # oracle on a seeded sample
    rng = np.random.default_rng(1)
    pick = sorted(rng.choice(b.n_windows, 48, replace=False).tolist())
    ref = oracle.consensus(b.select(pick), 3, -5, -4, True, 0)
    assert ref.consensus == [r1.consensus[i] for i in pick]
