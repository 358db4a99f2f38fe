"""The certificate of the exact banded DP (racon_amd/csrc/poa_band.hpp), on the CPU: the oracle's band study
(oracle/poa_oracle.cpp, Engine::band_study) evaluates every alignment once on full rows and once inside the kernel's window
policy (256 columns following the row's backbone coordinate in steps of 32), applies the certificate, and counts the
alignments whose certificate held although the banded result differed from the full one.  That count must be zero -- on
ordinary ONT-like windows, where nearly every alignment is certified, and on windows built to pull optimal paths off the
diagonal (tandem repeats, long indels, unrelated layers), where the certificate has to refuse."""
import ctypes as C

import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_windows
from test_gpu_band import long_window


def band_study(oracle, batch, scores, wb=256, g=32):
    lib = oracle.lib()
    lib.rcn_oracle_band_study.argtypes = [C.c_int, C.c_int]
    lib.rcn_oracle_band_study.restype = None
    lib.rcn_oracle_band_stats.argtypes = [C.POINTER(C.c_uint64)]
    lib.rcn_oracle_band_stats.restype = None
    lib.rcn_oracle_band_study(wb, g)
    try:
        oracle.consensus(batch, *scores, True, 1)          # one thread: the study's counters are process-wide
    finally:
        lib.rcn_oracle_band_study(0, 0)
    out = (C.c_uint64 * 10)()
    lib.rcn_oracle_band_stats(out)
    keys = ["alignments", "banded", "exact_cert_ok", "cheap_cert_ok", "same_result", "cert_ok_but_different", "max_alive_width",
            "alive_width_sum", "rows", "shifts"]
    st = dict(zip(keys, [int(v) for v in out]))
    print(st)
    return st


@pytest.mark.parametrize("scores", [(3, -5, -4), (5, -4, -8)])
def test_certificate_holds_and_is_sound_on_ont_like_windows(oracle, scores):
    b = simulate_windows(20_000, 500, 30.0, 10000, seed=11)
    st = band_study(oracle, b, scores)
    assert st["banded"] > 500
    assert st["cert_ok_but_different"] == 0
    assert st["exact_cert_ok"] >= 0.97 * st["banded"]          # "successors of alive cells are computed": a few per mille are redone
    assert st["same_result"] >= st["exact_cert_ok"]             # certified => identical (best sink and path); some uncertified ones are too


@pytest.mark.parametrize("seed", [1, 2])
def test_certificate_refuses_where_the_band_is_wrong(oracle, seed):
    rng = np.random.default_rng(seed)
    b = WindowBatch.from_windows([long_window(rng, s) for s in range(12)])
    st = band_study(oracle, b, (3, -5, -4))
    assert st["banded"] > 0
    assert st["cert_ok_but_different"] == 0
    assert st["exact_cert_ok"] < st["banded"]                   # these windows do defeat the band somewhere: the certificate says so
