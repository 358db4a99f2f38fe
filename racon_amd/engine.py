"""ctypes binding of the C ABI in include/racon_hip.h (racon_amd/csrc/libracon_hip.so).

There is no CPU fallback: `HipEngine()` raises if the library is missing or no
MI355X is visible.  This mirrors how racon's CUDAPolisher uses its batch object
(reference src/cuda/cudapolisher.cpp:228-333): create per device, feed windows,
generate consensus, read back per-window status.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .batch import ConsensusResult, RcnBatch, RcnResult, WindowBatch
from .layout import (CigarSet, OverlapSet, PairSet, RcnAlignStats, RcnBatchDims, RcnBuildStats, RcnCigarSet, RcnOverlapSet, RcnPairSet,
                     RcnReadSet, ReadSet)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RACON_HIP_LIB: another build of the same ABI (profiling / experiment variants under racon_amd/csrc/), as in the host layer
LIB_PATH = os.environ.get("RACON_HIP_LIB") or os.path.join(_HERE, "csrc", "libracon_hip.so")


class RcnEngineConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("match", C.c_int8), ("mismatch", C.c_int8), ("gap", C.c_int8),
                ("trim", C.c_uint8), ("arena_bytes", C.c_uint64), ("max_slots", C.c_uint32), ("flags", C.c_uint32)]


class RcnRunStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("n_launches", C.c_uint32), ("n_retried", C.c_uint32), ("dp_cells", C.c_uint64),
                ("dp_pred_cells", C.c_uint64), ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64),
                ("dp_bytes", C.c_uint64), ("phase_clocks", C.c_uint64 * 8), ("n_sink_ties", C.c_uint64),
                ("dp_cells_full", C.c_uint64), ("dp_bytes_full", C.c_uint64), ("n_banded", C.c_uint64), ("n_band_redone", C.c_uint64),
                ("band_redo_why", C.c_uint64 * 8), ("wg_per_cu", C.c_uint32), ("split_deep", C.c_uint32), ("split_cus", C.c_uint32),
                ("split_deep_per_cu", C.c_uint32), ("launch_ms", C.c_double * 2), ("n_code_wave", C.c_uint64),
                ("n_small", C.c_uint64), ("n_small_bailed", C.c_uint64), ("small_bail_why", C.c_uint64 * 9),
                ("small_work", C.c_uint64 * 6), ("launch_ms_mid", C.c_double), ("split_mid", C.c_uint32), ("split_mid_cus", C.c_uint32),
                ("split_mid_per_cu", C.c_uint32), ("reserved_", C.c_uint32)]


class RcnWindowDesc(C.Structure):
    _fields_ = [("type", C.c_uint8), ("n_seqs", C.c_uint32), ("seq", C.POINTER(C.c_char_p)),
                ("seq_len", C.POINTER(C.c_uint32)), ("qual", C.POINTER(C.c_char_p)),
                ("begin", C.POINTER(C.c_uint32)), ("end", C.POINTER(C.c_uint32))]


class RcnWindowRefs(C.Structure):
    _fields_ = [("n_windows", C.c_uint32), ("n_seqs", C.c_uint32), ("win_seq_off", C.c_void_p), ("win_type", C.c_void_p),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("seq_len", C.c_void_p), ("seq_begin", C.c_void_p), ("seq_end", C.c_void_p),
                ("flags", C.c_uint32)]


class RcnReserveHint(C.Structure):
    _fields_ = [("n_windows", C.c_uint32), ("n_seqs", C.c_uint32), ("n_bases", C.c_uint64), ("window_length", C.c_uint32),
                ("max_layer_length", C.c_uint32), ("max_window_bases", C.c_uint64)]


REFS_QUEUED = 1

EXPORTS = ["rcn_engine_create", "rcn_engine_destroy", "rcn_engine_upload", "rcn_engine_run", "rcn_engine_result",
           "rcn_engine_stats", "rcn_engine_set_trim", "rcn_engine_add_window", "rcn_engine_has_windows", "rcn_engine_generate_consensus",
           "rcn_engine_reset", "rcn_device_count", "rcn_strerror", "rcn_version",
           "rcn_engine_build_windows", "rcn_engine_build_windows_from_cigars", "rcn_engine_build_stats", "rcn_engine_batch_dims", "rcn_engine_export_batch", "rcn_engine_polish", "rcn_device_free_memory",
           "rcn_engine_align_pairs", "rcn_engine_alignment_cigars", "rcn_engine_align_stats", "rcn_engine_build_windows_from_pairs",
           "rcn_engine_polish_refs", "rcn_engine_reserve", "rcn_engine_reserve_refs", "rcn_engine_reserve_run", "rcn_engine_verify", "rcn_engine_forget"]

_lib = None


def load_library():
    """Loads libracon_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP engine)")
    # One HIP runtime per process: PyTorch-ROCm brings its own libamdhip64 (same SONAME as /opt/rocm's).  Whoever loads first
    # wins for both, and a process that loaded this library first and torch second was seen to end up without a usable device
    # (`python __graft_entry__.py smoke` = build() then smoke()).  torch first, always -- it is what bench.py does anyway.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.rcn_engine_create.argtypes = [C.POINTER(RcnEngineConfig), C.POINTER(C.c_void_p)]
    lib.rcn_engine_destroy.argtypes = [C.c_void_p]
    lib.rcn_engine_destroy.restype = None
    lib.rcn_engine_upload.argtypes = [C.c_void_p, C.POINTER(RcnBatch)]
    lib.rcn_engine_run.argtypes = [C.c_void_p]
    lib.rcn_engine_polish.argtypes = [C.c_void_p, C.POINTER(RcnBatch)]
    lib.rcn_engine_polish_refs.argtypes = [C.c_void_p, C.POINTER(RcnWindowRefs)]
    lib.rcn_engine_reserve.argtypes = [C.c_void_p, C.POINTER(RcnReserveHint)]
    lib.rcn_engine_reserve_refs.argtypes = [C.c_void_p, C.POINTER(RcnWindowRefs)]
    lib.rcn_engine_reserve_run.argtypes = [C.c_void_p]
    lib.rcn_device_free_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.rcn_engine_result.argtypes = [C.c_void_p, C.POINTER(RcnResult)]
    lib.rcn_engine_stats.argtypes = [C.c_void_p, C.POINTER(RcnRunStats)]
    lib.rcn_engine_set_trim.argtypes = [C.c_void_p, C.c_int]
    lib.rcn_engine_add_window.argtypes = [C.c_void_p, C.POINTER(RcnWindowDesc)]
    lib.rcn_engine_has_windows.argtypes = [C.c_void_p]
    lib.rcn_engine_generate_consensus.argtypes = [C.c_void_p]
    lib.rcn_engine_reset.argtypes = [C.c_void_p]
    lib.rcn_engine_build_windows.argtypes = [C.c_void_p, C.POINTER(RcnReadSet), C.POINTER(RcnOverlapSet), C.c_uint32, C.c_double, C.c_uint8]
    lib.rcn_engine_build_windows_from_cigars.argtypes = [C.c_void_p, C.POINTER(RcnReadSet), C.POINTER(RcnCigarSet), C.c_uint32, C.c_double, C.c_uint8]
    lib.rcn_engine_build_stats.argtypes = [C.c_void_p, C.POINTER(RcnBuildStats)]
    lib.rcn_engine_align_pairs.argtypes = [C.c_void_p, C.POINTER(RcnReadSet), C.POINTER(RcnPairSet)]
    lib.rcn_engine_alignment_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    lib.rcn_engine_align_stats.argtypes = [C.c_void_p, C.POINTER(RcnAlignStats)]
    lib.rcn_engine_build_windows_from_pairs.argtypes = [C.c_void_p, C.POINTER(RcnReadSet), C.POINTER(RcnPairSet), C.c_uint32, C.c_double, C.c_uint8]
    lib.rcn_engine_batch_dims.argtypes = [C.c_void_p, C.POINTER(RcnBatchDims)]
    lib.rcn_engine_export_batch.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    lib.rcn_strerror.restype = C.c_char_p
    lib.rcn_strerror.argtypes = [C.c_int]
    lib.rcn_version.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc < 0:
        raise RuntimeError(f"{what} failed: {load_library().rcn_strerror(rc).decode()} ({rc})")
    return rc


class HipEngine:
    """One engine = one device + one stream (like one CUDABatchProcessor)."""

    def __init__(self, match: int = 3, mismatch: int = -5, gap: int = -4, trim: bool = True, device: int = 0,
                 arena_bytes: int = 0, max_slots: int = 0):
        self.lib = load_library()
        if self.lib.rcn_device_count() <= 0:
            raise RuntimeError("no HIP device visible: the racon_amd engine has no CPU fallback")
        cfg = RcnEngineConfig(device, match, mismatch, gap, int(trim), arena_bytes, max_slots, 0)
        self.h = C.c_void_p()
        _check(self.lib.rcn_engine_create(C.byref(cfg), C.byref(self.h)), "rcn_engine_create")
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.rcn_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # whole-batch form ---------------------------------------------------------
    def upload(self, batch: WindowBatch):
        cb = batch.as_c()
        self._keep = (batch, cb)
        _check(self.lib.rcn_engine_upload(self.h, C.byref(cb)), "rcn_engine_upload")

    def run(self) -> ConsensusResult:
        _check(self.lib.rcn_engine_run(self.h), "rcn_engine_run")
        return self.result()

    def run_only(self):
        _check(self.lib.rcn_engine_run(self.h), "rcn_engine_run")

    def result(self) -> ConsensusResult:
        r = RcnResult()
        _check(self.lib.rcn_engine_result(self.h, C.byref(r)), "rcn_engine_result")
        return ConsensusResult.from_c(r)

    def verify(self, fraction: float = 1.0) -> dict:
        """rcn_engine_verify: a deterministic sample of the last run's windows again on the GPU with every shortcut rule off
        (full rows, score-matrix traceback, spoa's DFS order), compared with what the run returned."""
        class Rep(C.Structure):
            _fields_ = [("n_checked", C.c_uint32), ("n_differ", C.c_uint32), ("first_window", C.c_uint32), ("n_int32", C.c_uint32), ("ms", C.c_double)]
        r = Rep()
        self.lib.rcn_engine_verify.argtypes = [C.c_void_p, C.c_double, C.POINTER(Rep)]
        _check(self.lib.rcn_engine_verify(self.h, float(fraction), C.byref(r)), "rcn_engine_verify")
        return {k: getattr(r, k) for k, _ in Rep._fields_}

    def stats(self) -> dict:
        s = RcnRunStats()
        _check(self.lib.rcn_engine_stats(self.h, C.byref(s)), "rcn_engine_stats")
        d = {k: getattr(s, k) for k, _ in RcnRunStats._fields_}
        d["phase_clocks"] = list(s.phase_clocks)
        d["band_redo_why"] = list(s.band_redo_why)
        d["launch_ms"] = list(s.launch_ms)
        d["small_bail_why"] = list(s.small_bail_why)
        d["small_work"] = list(s.small_work)
        return d

    def consensus(self, batch: WindowBatch) -> ConsensusResult:
        """upload + run of one batch, the copy hidden behind the kernel (rcn_engine_polish)."""
        cb = batch.as_c()
        self._keep = (batch, cb)
        _check(self.lib.rcn_engine_polish(self.h, C.byref(cb)), "rcn_engine_polish")
        return self.result()

    def _refs(self, batch: WindowBatch, flags: int):
        batch.as_c()                                    # contiguous arrays of the ABI's dtypes
        ns = batch.n_seqs
        so = batch.seq_off.astype(np.uint64)
        seq = (np.uint64(batch.bases.ctypes.data) + so[:-1]).astype(np.uint64)
        qual = np.where(batch.seq_has_qual != 0, np.uint64(batch.quals.ctypes.data) + so[:-1], np.uint64(0)).astype(np.uint64)
        ln = (so[1:] - so[:-1]).astype(np.uint32)
        r = RcnWindowRefs(batch.n_windows, ns, batch.win_seq_off.ctypes.data, batch.win_type.ctypes.data, seq.ctypes.data, qual.ctypes.data,
                          ln.ctypes.data, batch.seq_begin.ctypes.data, batch.seq_end.ctypes.data, flags)
        self._keep = (batch, seq, qual, ln)
        return r

    def consensus_refs(self, batch: WindowBatch, flags: int = 0) -> ConsensusResult:
        """The same batch handed over as borrowed per-sequence pointers (rcn_engine_polish_refs: the form racon::Window holds
        its sequences in); sequences without quality get a NULL quality pointer."""
        r = self._refs(batch, flags)
        _check(self.lib.rcn_engine_polish_refs(self.h, C.byref(r)), "rcn_engine_polish_refs")
        return self.result()

    def reserve_refs(self, batch: WindowBatch, flags: int = 0):
        """Everything consensus_refs(batch) will allocate, ahead of time (rcn_engine_reserve_refs)."""
        r = self._refs(batch, flags)
        _check(self.lib.rcn_engine_reserve_refs(self.h, C.byref(r)), "rcn_engine_reserve_refs")

    def reserve(self, n_windows: int, n_seqs: int, n_bases: int, window_length: int, max_layer_length: int = 0, max_window_bases: int = 0):
        """Allocation ahead of the first batch (rcn_engine_reserve)."""
        h = RcnReserveHint(n_windows, n_seqs, n_bases, window_length, max_layer_length, max_window_bases)
        _check(self.lib.rcn_engine_reserve(self.h, C.byref(h)), "rcn_engine_reserve")

    # device-side window construction (reference src/polisher.cpp:388-461) --------
    def build_windows(self, reads: ReadSet, overlaps: OverlapSet, window_length: int, quality_threshold: float, window_type: int):
        """Builds every window of every target in HBM from the resident reads and breaking points; the batch stays
        resident for run()."""
        cr, co = reads.as_c(), overlaps.as_c()
        self._keep = (reads, overlaps, cr, co)
        _check(self.lib.rcn_engine_build_windows(self.h, C.byref(cr), C.byref(co), int(window_length), float(quality_threshold),
                                                 int(window_type)), "rcn_engine_build_windows")

    def build_windows_from_cigars(self, reads: ReadSet, alignments: CigarSet, window_length: int, quality_threshold: float, window_type: int):
        """The same from the alignments: breaking points (reference src/overlap.cpp:226-292) are also found on the device."""
        cr, ca = reads.as_c(), alignments.as_c()
        self._keep = (reads, alignments, cr, ca)
        _check(self.lib.rcn_engine_build_windows_from_cigars(self.h, C.byref(cr), C.byref(ca), int(window_length), float(quality_threshold),
                                                             int(window_type)), "rcn_engine_build_windows_from_cigars")

    # exact pairwise alignment on the device (reference src/overlap.cpp:205-224) ------
    def align_pairs(self, reads: ReadSet, pairs: PairSet):
        """Aligns every pair in HBM (the paths stay there); alignment_cigars() brings them back as CIGAR strings."""
        cr, cp = reads.as_c(), pairs.as_c()
        self._keep = (reads, pairs, cr, cp)
        self._n_pairs = pairs.n_pairs
        _check(self.lib.rcn_engine_align_pairs(self.h, C.byref(cr), C.byref(cp)), "rcn_engine_align_pairs")

    def alignment_cigars(self):
        """(list of CIGAR bytes, int32 array of edit distances) of the last align_pairs()."""
        n = self._n_pairs
        off = np.zeros(n + 1, np.uint64)
        dist = np.zeros(max(n, 1), np.int32)
        need = C.c_uint64(0)
        _check(self.lib.rcn_engine_alignment_cigars(self.h, off.ctypes.data_as(C.c_void_p), None, 0, C.byref(need), dist.ctypes.data_as(C.c_void_p)),
               "rcn_engine_alignment_cigars")
        buf = np.zeros(max(int(need.value), 1), np.uint8)
        _check(self.lib.rcn_engine_alignment_cigars(self.h, off.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), int(need.value),
                                                    C.byref(need), dist.ctypes.data_as(C.c_void_p)), "rcn_engine_alignment_cigars")
        return [buf[int(off[i]):int(off[i + 1])].tobytes() for i in range(n)], dist[:n]

    def align_stats(self) -> dict:
        s = RcnAlignStats()
        _check(self.lib.rcn_engine_align_stats(self.h, C.byref(s)), "rcn_engine_align_stats")
        return {k: getattr(s, k) for k, _ in RcnAlignStats._fields_}

    def build_windows_from_pairs(self, reads: ReadSet, pairs: PairSet, window_length: int, quality_threshold: float, window_type: int):
        """Alignment (edlib-equivalent, byte-identical paths), breaking points and window construction, all in HBM."""
        cr, cp = reads.as_c(), pairs.as_c()
        self._keep = (reads, pairs, cr, cp)
        self._n_pairs = pairs.n_pairs
        _check(self.lib.rcn_engine_build_windows_from_pairs(self.h, C.byref(cr), C.byref(cp), int(window_length), float(quality_threshold),
                                                            int(window_type)), "rcn_engine_build_windows_from_pairs")

    def build_stats(self) -> dict:
        s = RcnBuildStats()
        _check(self.lib.rcn_engine_build_stats(self.h, C.byref(s)), "rcn_engine_build_stats")
        return {k: getattr(s, k) for k, _ in RcnBuildStats._fields_}

    def export_batch(self) -> WindowBatch:
        """A host copy of the resident batch (uploaded or built on the device)."""
        d = RcnBatchDims()
        _check(self.lib.rcn_engine_batch_dims(self.h, C.byref(d)), "rcn_engine_batch_dims")
        nw, ns, nb = int(d.n_windows), int(d.n_seqs), int(d.n_bases)
        b = WindowBatch(np.zeros(nw + 1, np.uint32), np.zeros(nw, np.uint8), np.zeros(ns + 1, np.uint64), np.zeros(ns, np.uint8),
                        np.zeros(ns, np.uint32), np.zeros(ns, np.uint32), np.zeros(nb, np.uint8), np.zeros(nb, np.uint8))
        _check(self.lib.rcn_engine_export_batch(self.h, *[a.ctypes.data_as(C.c_void_p) for a in
               (b.win_seq_off, b.win_type, b.seq_off, b.seq_has_qual, b.seq_begin, b.seq_end, b.bases, b.quals)]), "rcn_engine_export_batch")
        return b

    # incremental form (CUDABatchProcessor::addWindow ...) ------------------------
    def add_window(self, window: dict) -> bool:
        seqs = window["seqs"]
        n = len(seqs)
        sp = (C.c_char_p * n)(*[bytes(s[0]) for s in seqs])
        qp = (C.c_char_p * n)(*[(bytes(s[1]) if s[1] is not None else None) for s in seqs])
        ln = (C.c_uint32 * n)(*[len(s[0]) for s in seqs])
        bg = (C.c_uint32 * n)(*[s[2] for s in seqs])
        en = (C.c_uint32 * n)(*[s[3] for s in seqs])
        d = RcnWindowDesc(int(window.get("type", 1)), n, sp, ln, qp, bg, en)
        rc = _check(self.lib.rcn_engine_add_window(self.h, C.byref(d)), "rcn_engine_add_window")
        return rc == 0

    def has_windows(self) -> bool:
        return bool(self.lib.rcn_engine_has_windows(self.h))

    def generate_consensus(self) -> ConsensusResult:
        _check(self.lib.rcn_engine_generate_consensus(self.h), "rcn_engine_generate_consensus")
        return self.result()

    def reset(self):
        _check(self.lib.rcn_engine_reset(self.h), "rcn_engine_reset")
