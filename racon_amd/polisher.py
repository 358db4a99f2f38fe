"""ctypes binding of the host layer (include/racon_host.h, racon_amd/host/libracon_host.so):
racon's Polisher surface — createPolisher / initialize / polish (reference
src/polisher.hpp:42-57) — plus the two halves of polish() (`windows()`,
`assemble()`) that let a consensus backend be swapped underneath for parity
checks.  The host layer is CPU code; `polish()` runs the consensus stage on the
MI355X and fails loudly when no device / libracon_hip.so is available.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .batch import ConsensusResult, RcnBatch, RcnResult, WindowBatch
from .layout import CigarSet, OverlapSet, PairSet, RcnCigarSet, RcnOverlapSet, RcnPairSet, RcnReadSet, ReadSet

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_DIR = os.path.join(_HERE, "host")
LIB_PATH = os.path.join(HOST_DIR, "libracon_host.so")


class RcnhParams(C.Structure):
    _fields_ = [("type", C.c_uint32), ("window_length", C.c_uint32), ("quality_threshold", C.c_double),
                ("error_threshold", C.c_double), ("trim", C.c_uint8), ("match", C.c_int8), ("mismatch", C.c_int8),
                ("gap", C.c_int8), ("num_threads", C.c_uint32), ("hip_batches", C.c_uint32)]


EXPORTS = ["rcnh_polisher_create", "rcnh_polisher_initialize", "rcnh_polisher_windows", "rcnh_polisher_assemble",
           "rcnh_polisher_polish", "rcnh_polisher_destroy", "rcnh_align_cigar", "rcnh_edit_distance", "rcnh_free",
           "rcnh_last_error", "rcnh_polisher_polish_seconds", "rcnh_polisher_num_windows", "rcnh_polisher_pairs",
           "rcnh_polisher_polish_plan"]

_lib = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", HOST_DIR, "all"])
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C racon_amd/host`")
    lib = C.CDLL(LIB_PATH)
    lib.rcnh_polisher_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(RcnhParams), C.POINTER(C.c_void_p)]
    lib.rcnh_polisher_initialize.argtypes = [C.c_void_p]
    lib.rcnh_polisher_windows.argtypes = [C.c_void_p, C.POINTER(RcnBatch)]
    lib.rcnh_polisher_keep_layout.argtypes = [C.c_void_p, C.c_int]
    lib.rcnh_polisher_layout.argtypes = [C.c_void_p, C.POINTER(RcnReadSet), C.POINTER(RcnOverlapSet), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.rcnh_polisher_alignments.argtypes = [C.c_void_p, C.POINTER(RcnCigarSet)]
    lib.rcnh_polisher_pairs.argtypes = [C.c_void_p, C.POINTER(RcnPairSet)]
    lib.rcnh_polisher_assemble.argtypes = [C.c_void_p, C.POINTER(RcnResult), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    lib.rcnh_polisher_polish.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    lib.rcnh_polisher_polish_seconds.argtypes = [C.c_void_p]
    lib.rcnh_polisher_polish_seconds.restype = C.c_double
    lib.rcnh_polisher_polish_plan.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rcnh_polisher_device_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rcnh_polisher_shard_input.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(RcnReadSet), C.POINTER(RcnOverlapSet),
                                              C.POINTER(RcnCigarSet), C.POINTER(RcnPairSet)]
    lib.rcnh_polisher_num_windows.argtypes = [C.c_void_p]
    lib.rcnh_polisher_num_windows.restype = C.c_uint64
    lib.rcnh_polisher_destroy.argtypes = [C.c_void_p]
    lib.rcnh_polisher_destroy.restype = None
    lib.rcnh_align_cigar.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.rcnh_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    lib.rcnh_edit_distance.restype = C.c_uint64
    lib.rcnh_free.argtypes = [C.c_void_p]
    lib.rcnh_free.restype = None
    lib.rcnh_last_error.restype = C.c_char_p
    _lib = lib
    return lib


class RaconError(RuntimeError):
    """A condition the reference reports with `[racon::...] error: ...` + exit(1)."""


def _check(rc):
    if rc != 0:
        raise RaconError(load_library().rcnh_last_error().decode(errors="replace"))


def _batch_from_c(cb: RcnBatch) -> WindowBatch:
    nw, ns = int(cb.n_windows), int(cb.n_seqs)

    def arr(p, n, dt):
        return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt, copy=True)

    seq_off = arr(cb.seq_off, ns + 1, np.uint64)
    nb = int(seq_off[-1]) if ns else 0
    return WindowBatch(arr(cb.win_seq_off, nw + 1, np.uint32), arr(cb.win_type, nw, np.uint8), seq_off,
                       arr(cb.seq_has_qual, ns, np.uint8), arr(cb.seq_begin, ns, np.uint32), arr(cb.seq_end, ns, np.uint32),
                       arr(cb.bases, nb, np.uint8), arr(cb.quals, nb, np.uint8))


def result_as_c(res: ConsensusResult):
    """ConsensusResult -> (RcnResult, keep-alive tuple)."""
    n = len(res.consensus)
    off = np.zeros(n + 1, np.uint64)
    if n:
        off[1:] = np.cumsum([len(c) for c in res.consensus], dtype=np.uint64)
    blob = np.frombuffer(b"".join(res.consensus) + b"\0", np.uint8).copy()
    pol = np.ascontiguousarray(res.polished, np.uint8) if n else np.zeros(1, np.uint8)
    chi = np.ascontiguousarray(res.chimeric, np.uint8) if n else np.zeros(1, np.uint8)
    r = RcnResult()
    r.n_windows = n
    r.cons_off = off.ctypes.data_as(C.POINTER(C.c_uint64))
    r.cons = blob.ctypes.data_as(C.POINTER(C.c_uint8))
    r.polished = pol.ctypes.data_as(C.POINTER(C.c_uint8))
    r.chimeric = chi.ctypes.data_as(C.POINTER(C.c_uint8))
    return r, (off, blob, pol, chi)


class Polisher:
    """racon::Polisher.  type: "kC" (contig polishing) or "kF" (fragment correction)."""

    def __init__(self, sequences_path: str, overlaps_path: str, target_path: str, type: str = "kC",
                 window_length: int = 500, quality_threshold: float = 10.0, error_threshold: float = 0.3,
                 trim: bool = True, match: int = 3, mismatch: int = -5, gap: int = -4, num_threads: int = 1,
                 hip_batches: int = 1):
        self.lib = load_library()
        t = {"kC": 0, "kF": 1}.get(type, type)
        q = RcnhParams(int(t), window_length, quality_threshold, error_threshold, int(trim), match, mismatch, gap,
                       num_threads, hip_batches)
        self.h = C.c_void_p()
        _check(self.lib.rcnh_polisher_create(os.fsencode(sequences_path), os.fsencode(overlaps_path),
                                             os.fsencode(target_path), C.byref(q), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.rcnh_polisher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initialize(self, keep_layout: bool = False):
        """keep_layout: also record the flattened sequences / overlaps (see layout())."""
        if keep_layout:
            _check(self.lib.rcnh_polisher_keep_layout(self.h, 1))
        _check(self.lib.rcnh_polisher_initialize(self.h))

    def layout(self):
        """(ReadSet, OverlapSet, window_type, window_length, quality_threshold): the input of the window construction
        (reference src/polisher.cpp:388-461) -- what HipEngine.build_windows / oracle.window_layout take."""
        r, o = RcnReadSet(), RcnOverlapSet()
        wt, wl, qt = C.c_uint8(), C.c_uint32(), C.c_double()
        _check(self.lib.rcnh_polisher_layout(self.h, C.byref(r), C.byref(o), C.byref(wt), C.byref(wl), C.byref(qt)))
        return ReadSet.from_c(r), OverlapSet.from_c(o), int(wt.value), int(wl.value), float(qt.value)

    def alignments(self) -> CigarSet:
        """The alignments (CIGAR + extents) the breaking points of layout() were derived from."""
        a = RcnCigarSet()
        _check(self.lib.rcnh_polisher_alignments(self.h, C.byref(a)))
        return CigarSet.from_c(a)

    def pairs(self) -> PairSet:
        """The segment pairs of the overlaps (what the pre-alignment of reference src/overlap.cpp:205-224 works on)."""
        a = RcnPairSet()
        _check(self.lib.rcnh_polisher_pairs(self.h, C.byref(a)))
        n = int(a.n_pairs)

        def arr(ptr, dt):
            return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True)
        return PairSet(arr(a.q_id, np.uint32), arr(a.t_id, np.uint32), arr(a.strand, np.uint8), arr(a.q_begin, np.uint32),
                       arr(a.q_end, np.uint32), arr(a.t_begin, np.uint32), arr(a.t_end, np.uint32))

    def device_plan(self, n_shards: int) -> dict:
        """The device-built path's cut of the job into `n_shards` window ranges (racon_amd/host/device_job.cpp): pure host code."""
        import numpy as np
        cut = np.zeros(n_shards + 1, np.uint64); lo = np.zeros(n_shards, np.uint64); hi = np.zeros(n_shards, np.uint64); no = np.zeros(n_shards, np.uint64)
        n = self.lib.rcnh_polisher_device_plan(self.h, n_shards, cut.ctypes.data, lo.ctypes.data, hi.ctypes.data, no.ctypes.data)
        if n < 0:
            _check(n)
        return {"n_shards": n, "cut": cut[:n + 1].astype(np.int64), "target_lo": lo[:n].astype(np.int64), "target_hi": hi[:n].astype(np.int64), "n_overlaps": no[:n].astype(np.int64)}

    def shard_input(self, n_shards: int, shard: int):
        """(dims, ReadSet, OverlapSet): one shard's input as its engine gets it -- its targets first, then the reads its overlaps point
        into, re-numbered; the engine's window l is the job's window dims["window_base"] + l."""
        import numpy as np
        dims = np.zeros(4, np.uint64)
        r, o = RcnReadSet(), RcnOverlapSet()
        _check(self.lib.rcnh_polisher_shard_input(self.h, n_shards, shard, dims.ctypes.data, C.byref(r), C.byref(o), None, None))
        d = dict(zip(("window_first", "window_last", "window_base", "n_windows_local"), (int(v) for v in dims)))
        return d, ReadSet.from_c(r), OverlapSet.from_c(o)

    def windows(self) -> WindowBatch:
        cb = RcnBatch()
        _check(self.lib.rcnh_polisher_windows(self.h, C.byref(cb)))
        return _batch_from_c(cb)

    def assemble(self, res: ConsensusResult, drop_unpolished_sequences: bool = True) -> bytes:
        r, keep = result_as_c(res)
        out = C.c_char_p()
        n = C.c_uint64()
        _check(self.lib.rcnh_polisher_assemble(self.h, C.byref(r), int(drop_unpolished_sequences), C.byref(out), C.byref(n)))
        del keep
        return C.string_at(out, n.value)

    def polish(self, drop_unpolished_sequences: bool = True) -> bytes:
        """FASTA text exactly as `racon` prints it (reference src/main.cpp:159-161)."""
        out = C.c_char_p()
        n = C.c_uint64()
        _check(self.lib.rcnh_polisher_polish(self.h, int(drop_unpolished_sequences), C.byref(out), C.byref(n)))
        return C.string_at(out, n.value)


    def polish_seconds(self) -> float:
        """The Logger-bracketed interval of the last polish() (reference src/polisher.cpp:493 -> :539-543)."""
        return float(self.lib.rcnh_polisher_polish_seconds(self.h))

    def polish_plan(self):
        """(chunks, engines that took at least one) of the last polish() on host-built windows."""
        c, e = C.c_uint32(), C.c_uint32()
        _check(self.lib.rcnh_polisher_polish_plan(self.h, C.byref(c), C.byref(e)))
        return int(c.value), int(e.value)

    def num_windows(self) -> int:
        return int(self.lib.rcnh_polisher_num_windows(self.h))


def align_cigar(query: bytes, target: bytes) -> str:
    lib = load_library()
    p = C.c_void_p()
    _check(lib.rcnh_align_cigar(query, len(query), target, len(target), C.byref(p)))
    s = C.string_at(p).decode()
    lib.rcnh_free(p)
    return s


def edit_distance(a: bytes, b: bytes) -> int:
    return int(load_library().rcnh_edit_distance(a, len(a), b, len(b)))


def parse_fasta(text: bytes):
    """[(header, sequence)] of FASTA text with one line per sequence."""
    lines = text.split(b"\n")
    return [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
