"""Seeded synthetic window sets for the configurations of BASELINE.json
(SURVEY.md §8(d)): a random contig, error-bearing reads laid over it, cut into
racon windows exactly as reference src/overlap.cpp:226-292 (CIGAR -> breaking
points) + src/polisher.cpp:405-458 (length / mean-quality filters, add_layer
coordinates) would cut them when the overlaps carry the simulator's true
alignment.  Output is a WindowBatch; there is no file I/O on this path.
"""
from __future__ import annotations

import numpy as np

from .batch import WindowBatch

_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _sim_read(rng, contig: np.ndarray, read_len: int, sub: float, ins: float, dele: float, with_quality: bool,
              phred_mean: float, phred_sd: float, phred_lo: int, phred_hi: int):
    """One error-bearing read over `contig` (the order of the rng calls is part of the seeded workloads' definition).
    Returns (ts, te, read, quality | None, deleted, has_ins, match columns, qpos: query index of every column's base)."""
    contig_len = len(contig)
    ts = int(rng.integers(0, max(1, contig_len - read_len // 4)))
    te = min(contig_len, ts + read_len)
    n = te - ts
    tgt = contig[ts:te]
    deleted = rng.random(n) < dele
    deleted[0] = deleted[-1] = False
    subst = rng.random(n) < sub
    base = tgt.copy()
    base[subst] = _ACGT[(np.searchsorted(_ACGT, tgt[subst]) + rng.integers(1, 4, int(subst.sum()))) % 4]
    has_ins = rng.random(n) < ins
    has_ins[-1] = False
    emit = (~deleted).astype(np.int64) + has_ins.astype(np.int64)      # bases emitted per target column
    qpos = np.concatenate([[0], np.cumsum(emit)])                        # query index of column's M base
    qlen = int(qpos[-1])
    read = np.empty(qlen, np.uint8)
    mcols = np.nonzero(~deleted)[0]
    read[qpos[mcols]] = base[mcols]
    icols = np.nonzero(has_ins)[0]
    read[qpos[icols] + (~deleted[icols]).astype(np.int64)] = _ACGT[rng.integers(0, 4, icols.size)]
    q = None
    if with_quality:
        q = np.clip(np.rint(rng.normal(phred_mean, phred_sd, qlen)), phred_lo, phred_hi).astype(np.uint8) + 33
    return ts, te, read, q, deleted, has_ins, mcols, qpos


def _low_complexity(contig: np.ndarray, share: float, n_rate: float, seed: int) -> np.ndarray:
    """Overwrite about `share` of a random contig with what real genomes hold and a uniform one never does: homopolymer
    runs, short tandem repeats (units of 2-6 bases), stretches over a two-letter alphabet -- alignments over these have many
    co-optimal paths, i.e. score ties at every level (predecessor choice, move priority, tied sinks) -- and, with `n_rate`,
    N bases (a fifth symbol: racon takes any byte, reference src/sequence.cpp:24-27).  A random stream of its own: the
    seeded workloads without these options are unchanged."""
    r = np.random.default_rng([seed, 0x10c0]); out = contig.copy(); n = len(out); pos = 0
    while pos < n:
        if r.random() >= share:
            pos += int(r.integers(10, 120)); continue
        kind = int(r.integers(0, 3))
        if kind == 0:
            ln = int(r.integers(3, 40)); out[pos:pos + ln] = _ACGT[r.integers(0, 4)]
        elif kind == 1:
            unit = _ACGT[r.integers(0, 4, int(r.integers(2, 7)))]; ln = len(unit) * int(r.integers(3, 30))
            out[pos:pos + ln] = np.resize(unit, ln)[:max(0, min(ln, n - pos))]
        else:
            two = _ACGT[r.choice(4, 2, replace=False)]; ln = int(r.integers(10, 150))
            out[pos:pos + ln] = two[r.integers(0, 2, ln)][:max(0, min(ln, n - pos))]
        pos += ln
    if n_rate > 0:
        out[r.random(n) < n_rate] = ord("N")
    return out


def simulate_windows(contig_len: int, window_len: int = 500, coverage: float = 30.0, read_len: int = 10000,
                     sub: float = 0.03, ins: float = 0.03, dele: float = 0.04, seed: int = 20260921,
                     phred_mean: float = 15.0, phred_sd: float = 4.0, phred_lo: int = 5, phred_hi: int = 30,
                     with_quality: bool = True, quality_threshold: float = 10.0, tgs: bool | None = None,
                     backbone_errors: float = 0.0, low_complexity: float = 0.0, n_rate: float = 0.0) -> WindowBatch:
    rng = np.random.default_rng(seed)
    contig = _ACGT[rng.integers(0, 4, contig_len)]
    if low_complexity > 0 or n_rate > 0:
        contig = _low_complexity(contig, low_complexity, n_rate, seed)
    # the backbone (draft assembly) may itself carry substitution errors
    backbone = contig.copy()
    if backbone_errors > 0:
        e = rng.random(contig_len) < backbone_errors
        backbone[e] = _ACGT[(np.searchsorted(_ACGT, backbone[e]) + rng.integers(1, 4, int(e.sum()))) % 4]

    n_win = (contig_len + window_len - 1) // window_len
    layers = [[] for _ in range(n_win)]          # (bases, qual|None, begin, end) in read (overlap-file) order
    n_reads = int(round(coverage * contig_len / read_len))
    total_read_len = 0
    for _ in range(n_reads):
        ts, te, read, q, deleted, has_ins, mcols, qpos = _sim_read(rng, contig, read_len, sub, ins, dele, with_quality,
                                                                   phred_mean, phred_sd, phred_lo, phred_hi)
        total_read_len += len(read)
        # cut at window boundaries: first / last MATCH column inside each window
        w0, w1 = ts // window_len, (te - 1) // window_len
        for w in range(w0, w1 + 1):
            a = max(ts, w * window_len) - ts
            b = min(te, (w + 1) * window_len) - ts
            m = mcols[(mcols >= a) & (mcols < b)]
            if m.size == 0:
                continue
            first_t, last_t = int(m[0]), int(m[-1])
            q0, q1 = int(qpos[first_t]), int(qpos[last_t]) + 1
            if (q1 - q0) < 0.02 * window_len:                      # polisher.cpp:415
                continue
            if with_quality:
                if float(np.mean(q[q0:q1].astype(np.float64) - 33.0)) < quality_threshold:   # polisher.cpp:419-433
                    continue
            begin = ts + first_t - w * window_len
            end = ts + last_t + 1 - w * window_len - 1             # polisher.cpp:454-457
            if begin == end:                                        # window.cpp:45-47
                continue
            layers[w].append((read[q0:q1].tobytes(), q[q0:q1].tobytes() if with_quality else None, begin, end))
    if tgs is None:
        tgs = (total_read_len / max(1, n_reads)) > 1000               # polisher.cpp:277-278
    windows = []
    for w in range(n_win):
        bb = backbone[w * window_len:min(contig_len, (w + 1) * window_len)].tobytes()
        windows.append({"type": 1 if tgs else 0, "seqs": [(bb, b"!" * len(bb), 0, 0)] + layers[w]})
    return WindowBatch.from_windows(windows)


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _cigar_of(n: int, deleted: np.ndarray, has_ins: np.ndarray) -> bytes:
    """The true alignment of a simulated read as CIGAR text: per target column D or M, an I behind a column with an insertion."""
    pos = np.arange(n) + np.cumsum(has_ins) - has_ins
    ops = np.full(n + int(has_ins.sum()), ord("I"), np.uint8)
    ops[pos] = np.where(deleted, ord("D"), ord("M"))
    cut = np.nonzero(np.diff(ops))[0] + 1
    starts = np.concatenate([[0], cut]); lens = np.diff(np.concatenate([starts, [ops.size]]))
    return b"".join(b"%d%c" % (int(l), int(ops[a])) for a, l in zip(starts.tolist(), lens.tolist()))


def _sim_piece_files(args):
    """One stretch of simulate_window_files as text: (target FASTA, reads FASTQ, SAM, PAF) of contig `index`."""
    contig_len, coverage, read_len, seed, index, sub, ins, dele, phred = args
    rng = np.random.default_rng(seed)                      # the stream of simulate_windows(contig_len, ..., seed)
    srng = np.random.default_rng([seed, 0x5bd1e995])       # strands: a stream of their own
    contig = _ACGT[rng.integers(0, 4, contig_len)]
    tname = b"ctg%d" % index
    fa = b">" + tname + b"\n" + contig.tobytes() + b"\n"
    fq, sam, paf = [], [b"@SQ\tSN:" + tname + b"\tLN:%d\n" % contig_len], []
    for i in range(int(round(coverage * contig_len / read_len))):
        ts, te, read, q, deleted, has_ins, mcols, qpos = _sim_read(rng, contig, read_len, sub, ins, dele, True, *phred)
        seq, qual = read.tobytes(), q.tobytes()
        strand = int(srng.integers(0, 2))
        name = b"r%d_%d" % (index, i)
        if strand:        # the read file holds the reverse complement; SAM lists SEQ on the target's strand
            fq.append(b"@" + name + b"\n" + seq.translate(_COMP)[::-1] + b"\n+\n" + qual[::-1] + b"\n")
        else:
            fq.append(b"@" + name + b"\n" + seq + b"\n+\n" + qual + b"\n")
        sam.append(name + b"\t%d\t" % (16 if strand else 0) + tname + b"\t%d\t60\t" % (ts + 1) + _cigar_of(te - ts, deleted, has_ins) +
                   b"\t*\t0\t0\t" + seq + b"\t" + qual + b"\n")
        paf.append(name + b"\t%d\t0\t%d\t" % (len(seq), len(seq)) + (b"-" if strand else b"+") + b"\t" + tname +
                   b"\t%d\t%d\t%d\t%d\t%d\t60\n" % (contig_len, ts, te, te - ts, te - ts))
    return fa, b"".join(fq), b"".join(sam), b"".join(paf)


def simulate_window_files(out_dir: str, contig_len: int, coverage: float = 30.0, read_len: int = 10000, seed: int = 20260921,
                          piece: int = 1_000_000, workers: int = 16, sub: float = 0.03, ins: float = 0.03, dele: float = 0.04,
                          phred=(15.0, 4.0, 5, 30)) -> dict:
    """The workload of simulate_windows_parallel(contig_len, ..., seed) as racon INPUT FILES (what `racon reads overlaps
    targets` takes, reference src/main.cpp:139-157): one target per `piece`-bp stretch (FASTA, the true contig: a backbone
    without quality, like simulate_windows'), the reads on both strands (FASTQ), and the read-to-target overlaps as SAM
    with the simulator's true CIGAR (reference src/overlap.cpp:192: no pre-alignment) and as PAF.  The windows the
    reference's initialize() cuts from these files (src/polisher.cpp:388-461) are, byte for byte and in the same order,
    the WindowBatch simulate_windows_parallel returns for the same arguments (tests/test_synth_files.py) -- so the product
    path (files -> initialize -> polish) and the kernel path (packed batch) can be timed and checked on ONE workload.
    Plain text, not gzip (the reference's parsers read both): these files only ever live in a scratch directory."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    sizes = [piece] * (contig_len // piece) + ([contig_len % piece] if contig_len % piece else [])
    seeds = [seed] if len(sizes) == 1 else [seed * 1000 + k for k in range(len(sizes))]
    jobs = [(n, coverage, read_len, sd, k, sub, ins, dele, tuple(phred)) for k, (n, sd) in enumerate(zip(sizes, seeds))]
    if len(jobs) > 1 and workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(max(1, min(workers, len(jobs)))) as pool:
            parts = pool.map(_sim_piece_files, jobs)
    else:
        parts = [_sim_piece_files(j) for j in jobs]
    paths = {k: os.path.join(out_dir, n) for k, n in (("targets", "targets.fasta"), ("reads", "reads.fastq"),
                                                       ("sam", "overlaps.sam"), ("paf", "overlaps.paf"))}
    for col, key in enumerate(("targets", "reads", "sam", "paf")):
        with open(paths[key], "wb") as f:
            if key == "sam":        # header lines first, then the records
                heads = [p[col][:p[col].index(b"\n") + 1] for p in parts]
                f.write(b"".join(heads))
                for p, h in zip(parts, heads):
                    f.write(p[col][len(h):])
            else:
                for p in parts:
                    f.write(p[col])
    return paths


def simulate_fragment_windows(genome_len: int, n_reads: int, read_len: int = 10000, window_len: int = 500,
                              sub: float = 0.03, ins: float = 0.03, dele: float = 0.04, seed: int = 20260924,
                              min_overlap: int = 2000, quality_threshold: float = 10.0,
                              phred_mean: float = 15.0, phred_sd: float = 4.0, phred_lo: int = 5, phred_hi: int = 30) -> WindowBatch:
    """Fragment correction (`racon -f`, BASELINE.json configs[4]; reference src/main.cpp:18-38, src/polisher.cpp:295 keeps
    every overlap per query in kF mode): the TARGETS are the reads themselves.  Error-bearing reads sampled from a random
    genome; every pair that shares at least `min_overlap` genome columns overlaps in BOTH directions (dual overlaps: A is a
    layer source for B's windows and B for A's); each target read is cut into windows of `window_len` of ITS coordinates
    and every overlapping read contributes the stretch between the first and the last genome column that carries a base
    in both reads inside the window -- what Overlap::find_breaking_points (reference src/overlap.cpp:226-292) derives from
    the pair's alignment when that alignment is the true one.  Backbones carry the read's own qualities (FASTQ targets:
    real backbone weights, reference src/polisher.cpp:396-399).  All reads are generated on the forward strand (the packed
    batch holds oriented bases either way)."""
    rng = np.random.default_rng(seed)
    genome = _ACGT[rng.integers(0, 4, genome_len)]
    starts = np.sort(rng.integers(0, max(1, genome_len - read_len // 2), n_reads))
    reads = []          # (ts, te, bases, quals, qpos[col] = read index of the column's base or -1)
    for ts in starts.tolist():
        te = min(genome_len, ts + read_len)
        n = te - ts
        tgt = genome[ts:te]
        deleted = rng.random(n) < dele
        deleted[0] = deleted[-1] = False
        subst = rng.random(n) < sub
        base = tgt.copy()
        base[subst] = _ACGT[(np.searchsorted(_ACGT, tgt[subst]) + rng.integers(1, 4, int(subst.sum()))) % 4]
        has_ins = rng.random(n) < ins
        has_ins[-1] = False
        emit = (~deleted).astype(np.int64) + has_ins.astype(np.int64)
        qstart = np.concatenate([[0], np.cumsum(emit)])
        qlen = int(qstart[-1])
        read = np.empty(qlen, np.uint8)
        mcols = np.nonzero(~deleted)[0]
        read[qstart[mcols]] = base[mcols]
        icols = np.nonzero(has_ins)[0]
        read[qstart[icols] + (~deleted[icols]).astype(np.int64)] = _ACGT[rng.integers(0, 4, icols.size)]
        q = np.clip(np.rint(rng.normal(phred_mean, phred_sd, qlen)), phred_lo, phred_hi).astype(np.uint8) + 33
        qpos = np.where(deleted, -1, qstart[:-1])
        reads.append((ts, te, read, q, qpos))
    windows = []
    lo = 0
    for a, (ats, ate, abases, aq, aqpos) in enumerate(reads):
        n_win = (len(abases) + window_len - 1) // window_len
        layers = [[] for _ in range(n_win)]
        while lo < n_reads and reads[lo][1] <= ats:
            lo += 1
        for b in range(lo, n_reads):
            bts, bte, bbases, bq, bqpos = reads[b]
            if bts >= ate:
                break
            if b == a:
                continue
            g0, g1 = max(ats, bts), min(ate, bte)
            if g1 - g0 < min_overlap:
                continue
            ca = aqpos[g0 - ats:g1 - ats]
            cb = bqpos[g0 - bts:g1 - bts]
            both = np.nonzero((ca >= 0) & (cb >= 0))[0]
            if both.size == 0:
                continue
            pa, pb = ca[both], cb[both]
            wid = pa // window_len
            cuts = np.nonzero(np.diff(wid))[0] + 1
            firsts = np.concatenate([[0], cuts]); lasts = np.concatenate([cuts - 1, [both.size - 1]])
            for f, l in zip(firsts.tolist(), lasts.tolist()):
                w = int(wid[f])
                q0, q1 = int(pb[f]), int(pb[l]) + 1
                if (q1 - q0) < 0.02 * window_len:
                    continue
                if float(np.mean(bq[q0:q1].astype(np.float64) - 33.0)) < quality_threshold:
                    continue
                begin, end = int(pa[f]) - w * window_len, int(pa[l]) - w * window_len
                if begin == end:
                    continue
                layers[w].append((bbases[q0:q1].tobytes(), bq[q0:q1].tobytes(), begin, end))
        for w in range(n_win):
            bb = abases[w * window_len:(w + 1) * window_len].tobytes()
            qq = aq[w * window_len:(w + 1) * window_len].tobytes()
            windows.append({"type": 1, "seqs": [(bb, qq, 0, 0)] + layers[w]})
    return WindowBatch.from_windows(windows)


def simulate_fragment_files(out_dir: str, genome_len: int, n_reads: int, read_len: int = 10000, sub: float = 0.03, ins: float = 0.03,
                            dele: float = 0.04, seed: int = 20260924, min_overlap: int = 2000,
                            phred=(15.0, 4.0, 5, 30)) -> dict:
    """The fragment-correction workload (BASELINE.json configs[4], `racon -f reads overlaps reads`) as INPUT FILES: the reads
    of simulate_fragment_windows(genome_len, n_reads, ..., seed) -- same rng stream, so the same reads -- on random strands
    (FASTQ), and the all-vs-all overlaps from the true coordinates as PAF, BOTH directions per pair (dual overlaps, grouped
    by query; reference src/polisher.cpp:295 keeps every overlap per query in kF mode).  PAF carries no alignment: the
    pre-alignment (reference src/overlap.cpp:205-224) is part of the job, on the host or on the device."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    srng = np.random.default_rng([seed, 0x5bd1e995])
    genome = _ACGT[rng.integers(0, 4, genome_len)]
    starts = np.sort(rng.integers(0, max(1, genome_len - read_len // 2), n_reads))
    reads = []          # (ts, te, read length, qstart[col] = read index of the first base emitted at or after the column, strand)
    paths = {"reads": os.path.join(out_dir, "reads.fastq"), "paf": os.path.join(out_dir, "overlaps.paf")}
    with open(paths["reads"], "wb") as fr:
        for i, ts in enumerate(starts.tolist()):
            te = min(genome_len, ts + read_len)
            n = te - ts
            tgt = genome[ts:te]
            deleted = rng.random(n) < dele
            deleted[0] = deleted[-1] = False
            subst = rng.random(n) < sub
            base = tgt.copy()
            base[subst] = _ACGT[(np.searchsorted(_ACGT, tgt[subst]) + rng.integers(1, 4, int(subst.sum()))) % 4]
            has_ins = rng.random(n) < ins
            has_ins[-1] = False
            emit = (~deleted).astype(np.int64) + has_ins.astype(np.int64)
            qstart = np.concatenate([[0], np.cumsum(emit)])
            qlen = int(qstart[-1])
            read = np.empty(qlen, np.uint8)
            mcols = np.nonzero(~deleted)[0]
            read[qstart[mcols]] = base[mcols]
            icols = np.nonzero(has_ins)[0]
            read[qstart[icols] + (~deleted[icols]).astype(np.int64)] = _ACGT[rng.integers(0, 4, icols.size)]
            q = np.clip(np.rint(rng.normal(phred[0], phred[1], qlen)), phred[2], phred[3]).astype(np.uint8) + 33
            strand = int(srng.integers(0, 2))
            seq, qual = read.tobytes(), q.tobytes()
            if strand:
                seq, qual = seq.translate(_COMP)[::-1], qual[::-1]
            fr.write(b"@f%d\n" % i + seq + b"\n+\n" + qual + b"\n")
            reads.append((ts, te, qlen, qstart.astype(np.int32), strand))      # (100 000 reads at full scale: 4 GB of these, not 8)
    n_ovl = 0
    with open(paths["paf"], "wb") as fp:
        lo = 0
        for a, (ats, ate, alen, aq, astr) in enumerate(reads):          # a = query
            while lo < n_reads and reads[lo][1] <= ats:
                lo += 1
            for b in range(lo, n_reads):
                bts, bte, blen, bq, bstr = reads[b]
                if bts >= ate:
                    break
                if b == a:
                    continue
                g0, g1 = max(ats, bts), min(ate, bte)
                if g1 - g0 < min_overlap:
                    continue
                qb, qe = int(aq[g0 - ats]), int(aq[g1 - ats])
                tb, te_ = int(bq[g0 - bts]), int(bq[g1 - bts])
                if qe - qb < 1 or te_ - tb < 1:
                    continue
                if astr:            # PAF query coordinates are on the query's own (file) strand
                    qb, qe = alen - qe, alen - qb
                if bstr:
                    tb, te_ = blen - te_, blen - tb
                fp.write(b"f%d\t%d\t%d\t%d\t%c\tf%d\t%d\t%d\t%d\t%d\t%d\t60\n" % (a, alen, qb, qe, b"-"[0] if astr != bstr else b"+"[0], b, blen, tb, te_,
                                                                                          min(qe - qb, te_ - tb), max(qe - qb, te_ - tb)))
                n_ovl += 1
    paths["n_overlaps"] = n_ovl
    return paths


def config_windows(name: str, scale: float = 1.0) -> WindowBatch:
    """The named synthetic configurations of SURVEY.md §8(d).  `scale` shrinks the
    contig (tests use small scales; bench.py uses 1.0)."""
    if name == "cfg2":      # 1 Mbp contig, 30x ONT-like reads, -w 500  (2000 windows at scale 1)
        return simulate_windows(int(1_000_000 * scale), 500, 30.0, 10000, seed=20260921)
    if name == "cfg3":      # 50 Mbp contig (8 GPUs)
        return simulate_windows(int(50_000_000 * scale), 500, 30.0, 10000, seed=20260922)
    if name == "cfg4":      # short reads: 150 bp at 60x, -w 200, kNGS
        return simulate_windows(int(1_000_000 * scale), 200, 60.0, 150, sub=0.003, ins=0.0005, dele=0.0005,
                                seed=20260923, phred_mean=30.0, phred_sd=0.0, phred_lo=30, phred_hi=30)
    if name == "ngs_w500":  # short reads on racon's default window: 150 bp at 60x, -w 500 (2000 windows of ~350 layers at scale 1)
        return simulate_windows(int(1_000_000 * scale), 500, 60.0, 150, sub=0.003, ins=0.0005, dele=0.0005,
                                seed=20260926, phred_mean=30.0, phred_sd=0.0, phred_lo=30, phred_hi=30)
    if name == "cfg5":      # fragment correction (-f): 100 000 reads x 10 kbp from a 33.3 Mbp genome, dual overlaps (2 M windows at scale 1)
        return simulate_fragment_windows(int(33_333_333 * scale), int(100_000 * scale), 10000, 500, seed=20260924)
    if name == "w1000":     # larger window (int32 score range)
        return simulate_windows(int(1_000_000 * scale), 1000, 30.0, 10000, seed=20260925)
    raise ValueError(name)


def simulate_files(out_dir: str, contig_len: int = 20000, coverage: float = 25.0, read_len: int = 3000,
                   sub: float = 0.03, ins: float = 0.03, dele: float = 0.04, backbone_errors: float = 0.03,
                   seed: int = 20260930, n_contigs: int = 1):
    """Writes a small polishing data set in racon's input formats (what `racon reads overlaps targets` takes,
    reference src/main.cpp:139-157): draft contigs (FASTA, with errors), reads (FASTQ, both strands), and the
    read-to-draft overlaps twice — SAM with the simulator's CIGAR (no pre-alignment needed, reference
    src/overlap.cpp:192) and PAF (pre-alignment on the host).  Returns the paths and the true contigs."""
    import gzip
    import os
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    os.makedirs(out_dir, exist_ok=True)
    paths = {k: os.path.join(out_dir, n) for k, n in (("targets", "draft.fasta.gz"), ("reads", "reads.fastq.gz"),
                                                       ("sam", "overlaps.sam.gz"), ("paf", "overlaps.paf.gz"))}
    truth = []
    with gzip.open(paths["targets"], "wb") as ft, gzip.open(paths["reads"], "wb") as fr, \
            gzip.open(paths["sam"], "wb") as fs, gzip.open(paths["paf"], "wb") as fp:
        rid = 0
        for ci in range(n_contigs):
            contig = _ACGT[rng.integers(0, 4, contig_len)]
            truth.append(contig.tobytes())
            # the draft: substitutions only, so that draft coordinates = true coordinates
            draft = contig.copy()
            e = rng.random(contig_len) < backbone_errors
            draft[e] = _ACGT[(np.searchsorted(_ACGT, draft[e]) + rng.integers(1, 4, int(e.sum()))) % 4]
            tname = b"contig%d" % ci
            ft.write(b">" + tname + b"\n" + draft.tobytes() + b"\n")
            fs.write(b"@SQ\tSN:" + tname + b"\tLN:%d\n" % contig_len)
            for _ in range(int(round(coverage * contig_len / read_len))):
                ts = int(rng.integers(0, max(1, contig_len - read_len // 2)))
                te = min(contig_len, ts + read_len)
                n = te - ts
                tgt = contig[ts:te]
                deleted = rng.random(n) < dele
                deleted[0] = deleted[-1] = False
                subst = rng.random(n) < sub
                base = tgt.copy()
                base[subst] = _ACGT[(np.searchsorted(_ACGT, tgt[subst]) + rng.integers(1, 4, int(subst.sum()))) % 4]
                has_ins = rng.random(n) < ins
                has_ins[-1] = False
                out = bytearray()
                ops = []                                   # CIGAR over the draft: M / I / D runs
                def push(op):
                    if ops and ops[-1][0] == op: ops[-1][1] += 1
                    else: ops.append([op, 1])
                for k in range(n):
                    if deleted[k]: push("D")
                    else: out.append(int(base[k])); push("M")
                    if has_ins[k]: out.append(int(_ACGT[rng.integers(0, 4)])); push("I")
                seq = bytes(out)
                qual = bytes((np.clip(np.rint(rng.normal(15, 4, len(seq))), 5, 30).astype(np.uint8) + 33).tolist())
                cigar = "".join("%d%s" % (c, o) for o, c in ops).encode()
                strand = int(rng.integers(0, 2))
                name = b"read%d" % rid
                rid += 1
                if strand:        # the read file holds the reverse complement; SAM lists SEQ on the forward strand
                    fr.write(b"@" + name + b"\n" + seq.translate(comp)[::-1] + b"\n+\n" + qual[::-1] + b"\n")
                else:
                    fr.write(b"@" + name + b"\n" + seq + b"\n+\n" + qual + b"\n")
                fs.write(name + b"\t%d\t" % (16 if strand else 0) + tname + b"\t%d\t60\t" % (ts + 1) + cigar +
                         b"\t*\t0\t0\t" + seq + b"\t" + qual + b"\n")
                fp.write(name + b"\t%d\t0\t%d\t" % (len(seq), len(seq)) + (b"-" if strand else b"+") + b"\t" + tname +
                         b"\t%d\t%d\t%d\t%d\t%d\t60\n" % (contig_len, ts, te, n, n))
    return paths, truth


def simulate_layout(contig_lens=(30000, 12345), window_len: int = 500, coverage: float = 20.0, read_len: int = 4000,
                    sub: float = 0.03, ins: float = 0.03, dele: float = 0.04, seed: int = 20260926,
                    frac_no_quality: float = 0.15, frac_low_quality: float = 0.1, target_quality: bool = False,
                    with_cigars: bool = False):
    """In-memory input of racon's window construction (racon_amd.layout.ReadSet / OverlapSet): draft targets, error-bearing
    reads on both strands (some without qualities, some with low ones so that the -q filter fires), and per overlap the
    breaking points Overlap::find_breaking_points would derive from the simulator's true alignment (first / last match of
    every window, reference src/overlap.cpp:226-292).  Returns (reads, overlaps, window_type)."""
    from .layout import CigarSet, OverlapSet, ReadSet
    rng = np.random.default_rng(seed)
    aligns = []
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    seqs, ovl = [], []
    targets = []
    for n in contig_lens:
        c = _ACGT[rng.integers(0, 4, n)]
        targets.append(c)
        tq = None
        if target_quality:
            tq = bytes((np.clip(np.rint(rng.normal(20, 5, n)), 2, 40).astype(np.uint8) + 33).tolist())
        seqs.append((c.tobytes(), tq))
    total_len = 0
    for ti, contig in enumerate(targets):
        n_t = len(contig)
        for _ in range(int(round(coverage * n_t / read_len))):
            ts = int(rng.integers(0, max(1, n_t - read_len // 3)))
            te = min(n_t, ts + int(rng.integers(read_len // 4, read_len + 1)))
            n = te - ts
            if n < 50:
                continue
            tgt = contig[ts:te]
            deleted = rng.random(n) < dele
            deleted[0] = deleted[-1] = False
            subst = rng.random(n) < sub
            base = tgt.copy()
            base[subst] = _ACGT[(np.searchsorted(_ACGT, tgt[subst]) + rng.integers(1, 4, int(subst.sum()))) % 4]
            has_ins = rng.random(n) < ins
            has_ins[-1] = False
            emit = (~deleted).astype(np.int64) + has_ins.astype(np.int64)
            qpos = np.concatenate([[0], np.cumsum(emit)])
            qlen = int(qpos[-1])
            read = np.empty(qlen, np.uint8)
            mcols = np.nonzero(~deleted)[0]
            read[qpos[mcols]] = base[mcols]
            icols = np.nonzero(has_ins)[0]
            read[qpos[icols] + (~deleted[icols]).astype(np.int64)] = _ACGT[rng.integers(0, 4, icols.size)]
            r = rng.random()
            if r < frac_no_quality:
                q = None
            else:
                mean = 6.0 if r < frac_no_quality + frac_low_quality else 15.0
                q = np.clip(np.rint(rng.normal(mean, 4, qlen)), 1, 30).astype(np.uint8) + 33
            # breaking points on the overlap's strand (the aligned, forward-on-target orientation)
            pts = []
            w0, w1 = ts // window_len, (te - 1) // window_len
            for w in range(w0, w1 + 1):
                a = max(ts, w * window_len) - ts
                b = min(te, (w + 1) * window_len) - ts
                m = mcols[(mcols >= a) & (mcols < b)]
                if m.size == 0:
                    continue
                pts.append((ts + int(m[0]), int(qpos[m[0]])))
                pts.append((ts + int(m[-1]) + 1, int(qpos[m[-1]]) + 1))
            strand = int(rng.integers(0, 2))
            rb = read.tobytes()
            qb = q.tobytes() if q is not None else None
            if strand:            # the read set holds the other strand; the overlap reads its reverse complement
                rb = rb.translate(comp)[::-1]
                qb = qb[::-1] if qb is not None else None
            if with_cigars:
                # the true alignment as a CIGAR: per target column D or M, an I behind the columns with an insertion
                pos = np.arange(n) + np.cumsum(has_ins) - has_ins
                ops = np.full(n + int(has_ins.sum()), ord("I"), np.uint8)
                ops[pos] = np.where(deleted, ord("D"), ord("M"))
                cut = np.nonzero(np.diff(ops))[0] + 1
                starts = np.concatenate([[0], cut]); lens = np.diff(np.concatenate([starts, [ops.size]]))
                cigar = b"".join(b"%d%c" % (int(l), int(ops[a])) for a, l in zip(starts, lens))
                aligns.append((len(seqs), ti, strand, 0, ts, te, cigar))
            ovl.append((len(seqs), ti, strand, pts))
            seqs.append((rb, qb))
            total_len += qlen
    n_reads = len(seqs) - len(targets)
    window_type = 1 if (total_len + sum(contig_lens)) / max(1, len(seqs)) > 1000 else 0
    if with_cigars:
        return ReadSet.from_sequences(seqs, len(targets)), OverlapSet.from_lists(ovl), window_type, CigarSet.from_lists(aligns)
    return ReadSet.from_sequences(seqs, len(targets)), OverlapSet.from_lists(ovl), window_type


def _sim_piece(args):
    contig_len, window_len, coverage, read_len, seed = args
    return simulate_windows(contig_len, window_len, coverage, read_len, seed=seed)


def simulate_windows_parallel(contig_len: int, window_len: int = 500, coverage: float = 30.0, read_len: int = 10000,
                              seed: int = 20260921, piece: int = 1_000_000, workers: int = 16) -> WindowBatch:
    """The windows of a long synthetic contig, generated as independent `piece`-bp stretches in worker processes (the
    generator is a Python loop per read: 8 s per Mbp) and concatenated: reads do not span stretches, everything else --
    depth distribution, error model, window shapes -- is the one-contig workload's.  A contig of at most one piece is
    exactly simulate_windows(contig_len, ..., seed)."""
    if contig_len <= piece:
        return simulate_windows(contig_len, window_len, coverage, read_len, seed=seed)
    sizes = [piece] * (contig_len // piece) + ([contig_len % piece] if contig_len % piece else [])
    jobs = [(n, window_len, coverage, read_len, seed * 1000 + k) for k, n in enumerate(sizes)]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(max(1, min(workers, len(jobs)))) as pool:
        parts = pool.map(_sim_piece, jobs)
    out = parts[0]
    for p in parts[1:]:
        out = out.concat(p)
    return out
