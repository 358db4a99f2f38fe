"""TEST INFRASTRUCTURE — CPU restatement of racon's window construction, the checker for
rcn_engine_build_windows (include/racon_hip.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this; the product path never does.

Restates, loop for loop, the end of Polisher::initialize:
  * reference src/polisher.cpp:388-403 — every target cut into windows of `window_length` (createWindow,
    src/window.cpp:15-40: backbone = element 0 with positions (0, 0); its quality is the target's or the dummy '!' string)
  * reference src/polisher.cpp:405-461 — every overlap, in order, cut at its breaking points: length filter (:415),
    mean-quality filter (:419-433), window rank / begin / end (:436-457), Window::add_layer (src/window.cpp:42-63);
    strand 1 reads the reverse complement (Sequence::create_reverse_complement, src/sequence.cpp:49-84)
Pinned: tests/test_window_layout.py compares it with the windows the host layer builds from the reference's own test
data (the host layer reproduces all 10 goldens of reference test/racon_test.cpp)."""
from __future__ import annotations

import numpy as np

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in ((65, 84), (84, 65), (67, 71), (71, 67)):        # A<->T, C<->G; everything else unchanged
    _COMP[_a] = _b


class LayoutError(ValueError):
    """[racon::Window::add_layer] error: layer begin and end positions are invalid!"""


def window_layout(reads, overlaps, window_length: int, quality_threshold: float, window_type: int):
    """reads: racon_amd.layout.ReadSet, overlaps: racon_amd.layout.OverlapSet -> racon_amd.batch.WindowBatch"""
    from racon_amd.batch import WindowBatch
    W = int(window_length)
    off = reads.seq_off.astype(np.int64)
    # ---- windows over every target (polisher.cpp:388-403)
    first_window = [0]
    windows = []
    for i in range(reads.n_targets):
        data = reads.bases[off[i]:off[i + 1]]
        qual = reads.quals[off[i]:off[i + 1]] if reads.seq_has_qual[i] else None
        k = 0
        for j in range(0, len(data), W):
            length = min(j + W, len(data)) - j
            bq = qual[j:j + length].tobytes() if qual is not None else b"!" * length
            windows.append({"type": int(window_type), "seqs": [(data[j:j + length].tobytes(), bq, 0, 0)]})
            k += 1
        first_window.append(first_window[-1] + k)
    # ---- layers (polisher.cpp:405-461), serial, in overlap order
    rc_cache = {}
    for o in range(overlaps.n_overlaps):
        qi, ti, rev = int(overlaps.q_id[o]), int(overlaps.t_id[o]), bool(overlaps.strand[o])
        a, z = int(off[qi]), int(off[qi + 1])
        if rev:
            if qi not in rc_cache:
                rc_cache[qi] = (_COMP[reads.bases[a:z][::-1]], reads.quals[a:z][::-1].copy())
            bases, quality = rc_cache[qi]
        else:
            bases, quality = reads.bases[a:z], reads.quals[a:z]
        has_quality = bool(reads.seq_has_qual[qi])
        p0, p1 = int(overlaps.bp_off[o]), int(overlaps.bp_off[o + 1])
        for j in range(p0, p1 - 1, 2):
            t0, t1 = int(overlaps.bp_t[j]), int(overlaps.bp_t[j + 1])
            q0, q1 = int(overlaps.bp_q[j]), int(overlaps.bp_q[j + 1])
            dl = (q1 - q0) & 0xFFFFFFFF                                   # uint32 arithmetic, as the reference
            if float(dl) < 0.02 * W:                                      # :415
                continue
            if has_quality:                                               # :419-433
                s = int(quality[q0:q1].astype(np.int64).sum()) - 33 * max(0, q1 - q0)
                if float(s) / float(dl) < quality_threshold:
                    continue
            window_rank = t0 // W
            window_start = window_rank * W
            begin = (t0 - window_start) & 0xFFFFFFFF
            end = (t1 - window_start - 1) & 0xFFFFFFFF
            # Window::add_layer (window.cpp:42-63)
            if dl == 0 or begin == end:
                continue
            n_win_t = first_window[ti + 1] - first_window[ti]
            if window_rank >= n_win_t:
                raise LayoutError("window rank outside the target")
            w = windows[first_window[ti] + window_rank]
            L = len(w["seqs"][0][0])
            if begin >= end or begin > L or end > L:
                raise LayoutError("[racon::Window::add_layer] error: layer begin and end positions are invalid!")
            w["seqs"].append((bases[q0:q1].tobytes(), quality[q0:q1].tobytes() if has_quality else None, begin, end))
    return WindowBatch.from_windows(windows)


def breaking_points(alignments, window_length: int):
    """CPU restatement of Overlap::breaking_points_from_cigar (reference src/overlap.cpp:226-292), base by base as there:
    racon_amd.layout.CigarSet -> racon_amd.layout.OverlapSet (first / last + 1 match column of every window touched)."""
    from racon_amd.layout import OverlapSet
    W = int(window_length)
    out = []
    text = alignments.cigar.tobytes()
    for o in range(alignments.n_overlaps):
        t_begin, t_end = int(alignments.t_begin[o]), int(alignments.t_end[o])
        ends = [i - 1 for i in range(0, t_end, W) if i > t_begin] + [t_end - 1]           # :228-236
        w, opened = 0, False
        first = last = (0, 0)
        q, t = int(alignments.q_start[o]) - 1, t_begin - 1                                  # :241-243
        pts = []
        n = 0
        for c in text[int(alignments.cigar_off[o]):int(alignments.cigar_off[o + 1])]:
            if 48 <= c <= 57:
                n = n * 10 + (c - 48)
                continue
            if c in (77, 61, 88):                                                           # M = X  (:248-267)
                for _ in range(n):
                    q += 1
                    t += 1
                    if not opened:
                        opened, first = True, (t, q)
                    last = (t + 1, q + 1)
                    if w < len(ends) and t == ends[w]:
                        if opened:
                            pts += [first, last]
                        opened = False
                        w += 1
            elif c == 73:                                                                   # I  (:268-270)
                q += n
            elif c in (68, 78):                                                             # D N  (:271-285)
                for _ in range(n):
                    t += 1
                    if w < len(ends) and t == ends[w]:
                        if opened:
                            pts += [first, last]
                        opened = False
                        w += 1
            n = 0
        out.append((int(alignments.q_id[o]), int(alignments.t_id[o]), int(alignments.strand[o]), pts))
    return OverlapSet.from_lists(out)
