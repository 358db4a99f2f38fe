"""The flattened input of racon's window construction (include/racon_hip.h: rcn_read_set / rcn_overlap_set): every
sequence of Polisher::sequences_ on its forward strand (targets first) and every kept overlap with its breaking
points — what the two loops of reference src/polisher.cpp:388-461 consume.  `HipEngine.build_windows` builds the
packed window batch from it in HBM; `oracle/window_layout.py` is the CPU restatement the tests compare with."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np


class RcnReadSet(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("n_targets", C.c_uint64), ("seq_off", C.POINTER(C.c_uint64)),
                ("bases", C.POINTER(C.c_uint8)), ("quals", C.POINTER(C.c_uint8)), ("seq_has_qual", C.POINTER(C.c_uint8))]


class RcnOverlapSet(C.Structure):
    _fields_ = [("n_overlaps", C.c_uint64), ("q_id", C.POINTER(C.c_uint32)), ("t_id", C.POINTER(C.c_uint32)),
                ("strand", C.POINTER(C.c_uint8)), ("bp_off", C.POINTER(C.c_uint64)), ("bp_t", C.POINTER(C.c_uint32)),
                ("bp_q", C.POINTER(C.c_uint32))]


class RcnCigarSet(C.Structure):
    _fields_ = [("n_overlaps", C.c_uint64), ("q_id", C.POINTER(C.c_uint32)), ("t_id", C.POINTER(C.c_uint32)),
                ("strand", C.POINTER(C.c_uint8)), ("q_start", C.POINTER(C.c_uint32)), ("t_begin", C.POINTER(C.c_uint32)),
                ("t_end", C.POINTER(C.c_uint32)), ("cigar_off", C.POINTER(C.c_uint64)), ("cigar", C.POINTER(C.c_uint8))]


class RcnPairSet(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("q_id", C.POINTER(C.c_uint32)), ("t_id", C.POINTER(C.c_uint32)),
                ("strand", C.POINTER(C.c_uint8)), ("q_begin", C.POINTER(C.c_uint32)), ("q_end", C.POINTER(C.c_uint32)),
                ("t_begin", C.POINTER(C.c_uint32)), ("t_end", C.POINTER(C.c_uint32))]


class RcnAlignStats(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("n_pairs", C.c_uint64), ("cells", C.c_uint64),
                ("ops_bytes", C.c_uint64), ("slots", C.c_uint32)]


class RcnBuildStats(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("gather_ms", C.c_double), ("n_pairs", C.c_uint64),
                ("n_layers", C.c_uint64), ("gather_bytes", C.c_uint64)]


class RcnBatchDims(C.Structure):
    _fields_ = [("n_windows", C.c_uint32), ("n_seqs", C.c_uint32), ("n_bases", C.c_uint64)]


def _ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(C.POINTER(ct))


@dataclass
class ReadSet:
    n_targets: int
    seq_off: np.ndarray        # uint64 [n_seqs + 1]
    bases: np.ndarray          # uint8, forward strand, upper case
    quals: np.ndarray          # uint8, phred+33 ('!' filler where a sequence has no quality)
    seq_has_qual: np.ndarray   # uint8 [n_seqs]

    @property
    def n_seqs(self) -> int:
        return int(self.seq_has_qual.shape[0])

    def as_c(self) -> RcnReadSet:
        for name, dt in (("seq_off", np.uint64), ("bases", np.uint8), ("quals", np.uint8), ("seq_has_qual", np.uint8)):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))
        return RcnReadSet(self.n_seqs, self.n_targets, _ptr(self.seq_off, C.c_uint64), _ptr(self.bases, C.c_uint8),
                          _ptr(self.quals, C.c_uint8), _ptr(self.seq_has_qual, C.c_uint8))

    @staticmethod
    def from_c(r: RcnReadSet) -> "ReadSet":
        n = int(r.n_seqs)
        off = np.ctypeslib.as_array(r.seq_off, shape=(n + 1,)).copy()
        nb = int(off[-1])
        return ReadSet(int(r.n_targets), off, np.ctypeslib.as_array(r.bases, shape=(max(nb, 1),))[:nb].copy(),
                       np.ctypeslib.as_array(r.quals, shape=(max(nb, 1),))[:nb].copy(),
                       np.ctypeslib.as_array(r.seq_has_qual, shape=(n,)).copy())

    @staticmethod
    def from_sequences(seqs, n_targets: int) -> "ReadSet":
        """seqs: [(bases: bytes, quality: bytes | None)], targets first."""
        off = np.zeros(len(seqs) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s, _ in seqs])
        bases = np.frombuffer(b"".join(s for s, _ in seqs), np.uint8).copy()
        quals = np.frombuffer(b"".join((q if q is not None else b"!" * len(s)) for s, q in seqs), np.uint8).copy()
        return ReadSet(n_targets, off, bases, quals, np.array([q is not None for _, q in seqs], np.uint8))


@dataclass
class OverlapSet:
    q_id: np.ndarray           # uint32 [n_overlaps]
    t_id: np.ndarray           # uint32
    strand: np.ndarray         # uint8
    bp_off: np.ndarray         # uint64 [n_overlaps + 1], in points
    bp_t: np.ndarray           # uint32 breaking_points_[k].first
    bp_q: np.ndarray           # uint32 breaking_points_[k].second

    @property
    def n_overlaps(self) -> int:
        return int(self.q_id.shape[0])

    def as_c(self) -> RcnOverlapSet:
        for name, dt in (("q_id", np.uint32), ("t_id", np.uint32), ("strand", np.uint8), ("bp_off", np.uint64),
                         ("bp_t", np.uint32), ("bp_q", np.uint32)):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))
        return RcnOverlapSet(self.n_overlaps, _ptr(self.q_id, C.c_uint32), _ptr(self.t_id, C.c_uint32), _ptr(self.strand, C.c_uint8),
                             _ptr(self.bp_off, C.c_uint64), _ptr(self.bp_t, C.c_uint32), _ptr(self.bp_q, C.c_uint32))

    @staticmethod
    def from_c(o: RcnOverlapSet) -> "OverlapSet":
        n = int(o.n_overlaps)
        off = np.ctypeslib.as_array(o.bp_off, shape=(n + 1,)).copy()
        npt = int(off[-1])
        # (an empty std::vector hands over a null pointer: no breaking points when they are left to the device)
        g = lambda p, k: np.ctypeslib.as_array(p, shape=(k,)).copy() if k and p else np.zeros(0, np.uint32)
        return OverlapSet(g(o.q_id, n), g(o.t_id, n), g(o.strand, n).astype(np.uint8), off, g(o.bp_t, npt), g(o.bp_q, npt))

    @staticmethod
    def from_lists(overlaps) -> "OverlapSet":
        """overlaps: [(q_id, t_id, strand, [(t_pos, q_pos), ...])]"""
        off = np.zeros(len(overlaps) + 1, np.uint64)
        off[1:] = np.cumsum([len(o[3]) for o in overlaps])
        pts = [p for o in overlaps for p in o[3]]
        return OverlapSet(np.array([o[0] for o in overlaps], np.uint32), np.array([o[1] for o in overlaps], np.uint32),
                          np.array([o[2] for o in overlaps], np.uint8), off,
                          np.array([p[0] for p in pts], np.uint32), np.array([p[1] for p in pts], np.uint32))


@dataclass
class CigarSet:
    """The alignments of the kept overlaps (include/racon_hip.h: rcn_cigar_set): what Overlap::find_breaking_points walks
    (reference src/overlap.cpp:226-292)."""
    q_id: np.ndarray           # uint32 [n_overlaps]
    t_id: np.ndarray
    strand: np.ndarray         # uint8
    q_start: np.ndarray        # uint32: first query position on the overlap's strand
    t_begin: np.ndarray        # uint32
    t_end: np.ndarray          # uint32
    cigar_off: np.ndarray      # uint64 [n_overlaps + 1]
    cigar: np.ndarray          # uint8 CIGAR text

    @property
    def n_overlaps(self) -> int:
        return int(self.q_id.shape[0])

    def as_c(self) -> RcnCigarSet:
        for name, dt in (("q_id", np.uint32), ("t_id", np.uint32), ("strand", np.uint8), ("q_start", np.uint32), ("t_begin", np.uint32),
                         ("t_end", np.uint32), ("cigar_off", np.uint64), ("cigar", np.uint8)):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))
        return RcnCigarSet(self.n_overlaps, _ptr(self.q_id, C.c_uint32), _ptr(self.t_id, C.c_uint32), _ptr(self.strand, C.c_uint8),
                           _ptr(self.q_start, C.c_uint32), _ptr(self.t_begin, C.c_uint32), _ptr(self.t_end, C.c_uint32),
                           _ptr(self.cigar_off, C.c_uint64), _ptr(self.cigar, C.c_uint8))

    @staticmethod
    def from_c(a: RcnCigarSet) -> "CigarSet":
        n = int(a.n_overlaps)
        off = np.ctypeslib.as_array(a.cigar_off, shape=(n + 1,)).copy()
        nb = int(off[-1])
        g = lambda p, k: np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].copy()
        return CigarSet(g(a.q_id, n), g(a.t_id, n), g(a.strand, n), g(a.q_start, n), g(a.t_begin, n), g(a.t_end, n), off, g(a.cigar, nb))

    @staticmethod
    def from_lists(al) -> "CigarSet":
        """al: [(q_id, t_id, strand, q_start, t_begin, t_end, cigar: bytes)]"""
        off = np.zeros(len(al) + 1, np.uint64)
        off[1:] = np.cumsum([len(a[6]) for a in al])
        col = lambda k, dt: np.array([a[k] for a in al], dt)
        return CigarSet(col(0, np.uint32), col(1, np.uint32), col(2, np.uint8), col(3, np.uint32), col(4, np.uint32), col(5, np.uint32),
                        off, np.frombuffer(b"".join(a[6] for a in al), np.uint8).copy())


@dataclass
class PairSet:
    """Overlaps that come without an alignment (include/racon_hip.h: rcn_pair_set): what Overlap::find_breaking_points
    hands to edlib (reference src/overlap.cpp:205-224) -- the query segment on the FORWARD read (reverse-complemented by
    the aligner when strand = 1) and the target segment."""
    q_id: np.ndarray           # uint32 [n_pairs]
    t_id: np.ndarray
    strand: np.ndarray         # uint8
    q_begin: np.ndarray        # uint32
    q_end: np.ndarray
    t_begin: np.ndarray
    t_end: np.ndarray

    @property
    def n_pairs(self) -> int:
        return int(self.q_id.shape[0])

    def as_c(self) -> RcnPairSet:
        for name, dt in (("q_id", np.uint32), ("t_id", np.uint32), ("strand", np.uint8), ("q_begin", np.uint32), ("q_end", np.uint32),
                         ("t_begin", np.uint32), ("t_end", np.uint32)):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))
        return RcnPairSet(self.n_pairs, _ptr(self.q_id, C.c_uint32), _ptr(self.t_id, C.c_uint32), _ptr(self.strand, C.c_uint8),
                          _ptr(self.q_begin, C.c_uint32), _ptr(self.q_end, C.c_uint32), _ptr(self.t_begin, C.c_uint32), _ptr(self.t_end, C.c_uint32))

    @staticmethod
    def from_lists(pairs) -> "PairSet":
        """pairs: [(q_id, t_id, strand, q_begin, q_end, t_begin, t_end)]"""
        col = lambda k, dt: np.array([p[k] for p in pairs], dt)
        return PairSet(col(0, np.uint32), col(1, np.uint32), col(2, np.uint8), col(3, np.uint32), col(4, np.uint32), col(5, np.uint32), col(6, np.uint32))
