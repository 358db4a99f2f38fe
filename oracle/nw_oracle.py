"""TEST INFRASTRUCTURE (oracle): the pairwise alignment of racon's breakpoint pre-alignment on the CPU.

What it restates: edlibAlign(q, ql, t, tl, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0)) +
edlibAlignmentToCigar(..., EDLIB_CIGAR_STANDARD) as called at reference src/overlap.cpp:205-224.  martinsos/edlib
v1.2.7 is an un-vendored dependency of the reference (CMakeLists.txt:43-48): the path-selection rule is the one
SURVEY.md Appendix B gives (validated there against the eight PAF / MHAP goldens of reference test/racon_test.cpp);
`oracle/spec_py.py` holds that validated code (col_scores / traceback / obtain) and this module only adds the CIGAR
encoding and the reverse complement of `Sequence::create_reverse_complement` (reference src/sequence.cpp:49-84).
Pinned by tests/test_pair_align_oracle.py against the host layer's aligner, which itself reproduces the PAF / MHAP
goldens end to end (tests/test_reference_goldens.py).  Only tests/ may import this."""
from __future__ import annotations

import numpy as np

from . import spec_py

_COMP = np.arange(256, dtype=np.uint8)
for a, b in ((ord("A"), ord("T")), (ord("C"), ord("G"))):
    _COMP[a], _COMP[b] = b, a


def reverse_complement(s: bytes) -> bytes:
    return _COMP[np.frombuffer(s, np.uint8)][::-1].tobytes()


def ops(query: bytes, target: bytes):
    """('M'|'I'|'D' list, edit distance) of query (rows) against target (columns)."""
    q, t = np.frombuffer(query, np.uint8), np.frombuffer(target, np.uint8)
    best = int(spec_py.col_scores(q, t)[-1]) if len(q) and len(t) else max(len(q), len(t))
    return spec_py.obtain(q, t, best), best


def rle(op_list) -> bytes:
    out, k = [], 0
    while k < len(op_list):
        j = k
        while j < len(op_list) and op_list[j] == op_list[k]:
            j += 1
        out.append(b"%d%s" % (j - k, op_list[k].encode()))
        k = j
    return b"".join(out)


def cigar(query: bytes, target: bytes):
    o, d = ops(query, target)
    return rle(o), d
