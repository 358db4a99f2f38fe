"""Packed window batches (the `rcn_batch` of include/racon_hip.h) on the Python side.

A WindowBatch holds, flattened, exactly what a list of racon::Window objects
holds (reference src/window.hpp:64-73): per window the backbone + layers
(`sequences_`, `qualities_`, `positions_`) and the window type.  numpy arrays
only; `as_c()` yields the ctypes struct that crosses the C ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np


class RcnBatch(C.Structure):
    _fields_ = [
        ("n_windows", C.c_uint32),
        ("n_seqs", C.c_uint32),
        ("win_seq_off", C.POINTER(C.c_uint32)),
        ("win_type", C.POINTER(C.c_uint8)),
        ("seq_off", C.POINTER(C.c_uint64)),
        ("seq_has_qual", C.POINTER(C.c_uint8)),
        ("seq_begin", C.POINTER(C.c_uint32)),
        ("seq_end", C.POINTER(C.c_uint32)),
        ("bases", C.POINTER(C.c_uint8)),
        ("quals", C.POINTER(C.c_uint8)),
    ]


class RcnResult(C.Structure):
    _fields_ = [
        ("n_windows", C.c_uint32),
        ("cons_off", C.POINTER(C.c_uint64)),
        ("cons", C.POINTER(C.c_uint8)),
        ("polished", C.POINTER(C.c_uint8)),
        ("chimeric", C.POINTER(C.c_uint8)),
    ]


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class WindowBatch:
    win_seq_off: np.ndarray   # uint32 [n_windows+1]
    win_type: np.ndarray      # uint8  [n_windows]
    seq_off: np.ndarray       # uint64 [n_seqs+1]
    seq_has_qual: np.ndarray  # uint8  [n_seqs]
    seq_begin: np.ndarray     # uint32 [n_seqs]
    seq_end: np.ndarray       # uint32 [n_seqs]
    bases: np.ndarray         # uint8
    quals: np.ndarray         # uint8

    @property
    def n_windows(self) -> int:
        return int(self.win_type.shape[0])

    @property
    def n_seqs(self) -> int:
        return int(self.seq_has_qual.shape[0])

    def as_c(self) -> RcnBatch:
        for name, dt in (("win_seq_off", np.uint32), ("win_type", np.uint8), ("seq_off", np.uint64),
                         ("seq_has_qual", np.uint8), ("seq_begin", np.uint32), ("seq_end", np.uint32),
                         ("bases", np.uint8), ("quals", np.uint8)):
            a = np.ascontiguousarray(getattr(self, name), dtype=dt)
            setattr(self, name, a)
        b = RcnBatch()
        b.n_windows = self.n_windows
        b.n_seqs = self.n_seqs
        b.win_seq_off = _ptr(self.win_seq_off, C.c_uint32)
        b.win_type = _ptr(self.win_type, C.c_uint8)
        b.seq_off = _ptr(self.seq_off, C.c_uint64)
        b.seq_has_qual = _ptr(self.seq_has_qual, C.c_uint8)
        b.seq_begin = _ptr(self.seq_begin, C.c_uint32)
        b.seq_end = _ptr(self.seq_end, C.c_uint32)
        b.bases = _ptr(self.bases, C.c_uint8)
        b.quals = _ptr(self.quals, C.c_uint8)
        return b

    # ---- construction helpers -------------------------------------------------
    @staticmethod
    def from_windows(windows: Sequence[dict]) -> "WindowBatch":
        """windows: [{'type': 0|1, 'seqs': [(bases: bytes, qual: bytes|None, begin, end), ...]}]
        with element 0 of 'seqs' the backbone (begin = end = 0)."""
        win_off = [0]
        wtype, soff, hq, bg, en = [], [0], [], [], []
        bases: List[bytes] = []
        quals: List[bytes] = []
        for w in windows:
            wtype.append(int(w.get("type", 1)))
            for (s, q, b, e) in w["seqs"]:
                bases.append(bytes(s))
                if q is None:
                    quals.append(b"!" * len(s))
                    hq.append(0)
                else:
                    assert len(q) == len(s)
                    quals.append(bytes(q))
                    hq.append(1)
                bg.append(b)
                en.append(e)
                soff.append(soff[-1] + len(s))
            win_off.append(len(hq))
        return WindowBatch(
            np.asarray(win_off, np.uint32), np.asarray(wtype, np.uint8), np.asarray(soff, np.uint64),
            np.asarray(hq, np.uint8), np.asarray(bg, np.uint32), np.asarray(en, np.uint32),
            np.frombuffer(b"".join(bases), np.uint8).copy() if bases else np.zeros(0, np.uint8),
            np.frombuffer(b"".join(quals), np.uint8).copy() if quals else np.zeros(0, np.uint8))

    def window(self, w: int) -> dict:
        s0, s1 = int(self.win_seq_off[w]), int(self.win_seq_off[w + 1])
        seqs = []
        for s in range(s0, s1):
            a, b = int(self.seq_off[s]), int(self.seq_off[s + 1])
            q = self.quals[a:b].tobytes() if self.seq_has_qual[s] else None
            seqs.append((self.bases[a:b].tobytes(), q, int(self.seq_begin[s]), int(self.seq_end[s])))
        return {"type": int(self.win_type[w]), "seqs": seqs}

    def select(self, idx: Sequence[int]) -> "WindowBatch":
        return WindowBatch.from_windows([self.window(int(i)) for i in idx])

    def concat(self, other: "WindowBatch") -> "WindowBatch":
        return WindowBatch(
            np.concatenate([self.win_seq_off, other.win_seq_off[1:] + self.win_seq_off[-1]]).astype(np.uint32),
            np.concatenate([self.win_type, other.win_type]),
            np.concatenate([self.seq_off, other.seq_off[1:] + self.seq_off[-1]]).astype(np.uint64),
            np.concatenate([self.seq_has_qual, other.seq_has_qual]),
            np.concatenate([self.seq_begin, other.seq_begin]),
            np.concatenate([self.seq_end, other.seq_end]),
            np.concatenate([self.bases, other.bases]),
            np.concatenate([self.quals, other.quals]))

    @staticmethod
    def window_costs(win_seq_off: np.ndarray, seq_off: np.ndarray) -> np.ndarray:
        """Cost proxy of a window for load balancing: (sequences) x (bases) -- the graph a layer is aligned against grows
        with every layer, so the DP work of a window goes with layers x bases, not with bases (the engine orders its
        work queue by the same proxy).  Needs the offset arrays only."""
        wso = win_seq_off.astype(np.int64)
        so = seq_off.astype(np.int64)
        bases = so[wso[1:]] - so[wso[:-1]]
        return (wso[1:] - wso[:-1]).astype(np.float64) * bases.astype(np.float64) + 1.0

    @staticmethod
    def shard_bounds(win_seq_off: np.ndarray, seq_off: np.ndarray, world: int) -> list:
        """world + 1 window indices: rank r holds windows [bounds[r], bounds[r + 1]) -- contiguous, balanced by window_costs."""
        n = len(win_seq_off) - 1
        if n <= 0:
            return [0] * (world + 1)
        cum = np.cumsum(WindowBatch.window_costs(win_seq_off, seq_off))
        bounds = [0]
        for r in range(1, world):
            bounds.append(max(bounds[-1], int(np.searchsorted(cum, cum[-1] * r / world, side="left"))))
        bounds.append(n)
        return bounds

    def shard(self, rank: int, world: int) -> Tuple["WindowBatch", np.ndarray]:
        """Contiguous, cost-balanced shard of the window index space for rank
        `rank` of `world` (windows are independent, reference
        src/polisher.cpp:496-503).  Returns the sub-batch and the global window
        indices it holds."""
        bounds = self.shard_bounds(self.win_seq_off, self.seq_off, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        idx = np.arange(lo, hi)
        s0, s1 = int(self.win_seq_off[lo]), int(self.win_seq_off[hi])
        b0, b1 = int(self.seq_off[s0]), int(self.seq_off[s1])
        sub = WindowBatch(
            (self.win_seq_off[lo:hi + 1] - self.win_seq_off[lo]).astype(np.uint32),
            self.win_type[lo:hi].copy(),
            (self.seq_off[s0:s1 + 1] - self.seq_off[s0]).astype(np.uint64),
            self.seq_has_qual[s0:s1].copy(), self.seq_begin[s0:s1].copy(), self.seq_end[s0:s1].copy(),
            self.bases[b0:b1].copy(), self.quals[b0:b1].copy())
        return sub, idx

    # ---- (de)serialisation: the dump format of tests/golden -----------------
    def save(self, path: str) -> None:
        np.savez_compressed(path, win_seq_off=self.win_seq_off, win_type=self.win_type, seq_off=self.seq_off,
                            seq_has_qual=self.seq_has_qual, seq_begin=self.seq_begin, seq_end=self.seq_end,
                            bases=self.bases, quals=self.quals)

    @staticmethod
    def load(path: str) -> "WindowBatch":
        z = np.load(path)
        return WindowBatch(z["win_seq_off"], z["win_type"], z["seq_off"], z["seq_has_qual"], z["seq_begin"],
                           z["seq_end"], z["bases"], z["quals"])


@dataclass
class ConsensusResult:
    consensus: List[bytes]
    polished: np.ndarray
    chimeric: np.ndarray

    @staticmethod
    def from_c(r: RcnResult) -> "ConsensusResult":
        n = int(r.n_windows)
        off = np.ctypeslib.as_array(r.cons_off, shape=(n + 1,)).copy() if n else np.zeros(1, np.uint64)
        total = int(off[-1])
        blob = bytes(np.ctypeslib.as_array(r.cons, shape=(max(total, 1),))[:total].tobytes()) if total else b""
        cons = [blob[int(off[i]):int(off[i + 1])] for i in range(n)]
        pol = np.ctypeslib.as_array(r.polished, shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        chi = np.ctypeslib.as_array(r.chimeric, shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        return ConsensusResult(cons, pol, chi)
