"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from racon_amd.batch import ConsensusResult, RcnBatch, RcnResult, WindowBatch  # noqa: E402

_lib = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "poa_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.rcn_oracle_consensus.restype = C.c_int
        _lib.rcn_oracle_consensus.argtypes = [C.POINTER(RcnBatch), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(RcnResult), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _lib.rcn_oracle_consensus_simd.restype = C.c_int
        _lib.rcn_oracle_consensus_simd.argtypes = _lib.rcn_oracle_consensus.argtypes
        _lib.rcn_oracle_free.argtypes = [C.c_void_p]
        _lib.rcn_oracle_tie_stats.argtypes = [C.POINTER(C.c_uint64)]
        _lib.rcn_oracle_tie_stats.restype = None
        _lib.rcn_oracle_edit_distance.restype = C.c_uint64
        _lib.rcn_oracle_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    return _lib


def consensus(batch: WindowBatch, m: int, x: int, g: int, trim: bool = True, threads: int = 0,
              with_stats: bool = False, simd: bool = False):
    """Oracle consensus of every window.  Returns ConsensusResult (and, with
    with_stats, per-window (cells, cells*(1+E/V)) arrays of SURVEY §8(d))."""
    if threads <= 0:
        threads = os.cpu_count() or 1
    cb = batch.as_c()
    res = RcnResult()
    h = C.c_void_p()
    n = batch.n_windows
    cells = np.zeros(max(n, 1), np.uint64)
    cxp = np.zeros(max(n, 1), np.float64)
    fn = lib().rcn_oracle_consensus_simd if simd else lib().rcn_oracle_consensus
    rc = fn(C.byref(cb), m, x, g, int(trim), threads, C.byref(res), C.byref(h),
                                    cells.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    cxp.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    out = ConsensusResult.from_c(res)
    lib().rcn_oracle_free(h)
    if with_stats:
        return out, cells[:n], cxp[:n]
    return out


def edit_distance(a: bytes, b: bytes) -> int:
    return int(lib().rcn_oracle_edit_distance(a, len(a), b, len(b)))


def tie_stats():
    """(sink-tie events, events the HIP kernel's rule classifies, of those rule == exact) since the last call."""
    a = (C.c_uint64 * 3)()
    lib().rcn_oracle_tie_stats(a)
    return int(a[0]), int(a[1]), int(a[2])
