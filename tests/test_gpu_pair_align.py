"""-m gpu: the device pairwise aligner (racon_amd/csrc/pair_align.hpp; rcn_engine_align_pairs /
rcn_engine_build_windows_from_pairs) against the oracle (oracle/nw_oracle.py, SURVEY Appendix B's rule) and the host
layer's aligner (racon_amd/host/nw_path.cpp, pinned by the reference's PAF / MHAP goldens): the CIGAR of every pair
must be identical byte for byte -- small pairs (plain traceback), pairs above the 1 MiB traceback-state threshold
(Hirschberg), both strands, degenerate shapes, alphabets beyond seven symbols (8-plane path), and the overlaps of the
reference's own PAF sample; the windows built from device alignments must equal the ones built from host CIGARs."""
import os

import numpy as np
import pytest

from helpers import REFDATA
from pairgen import mutate, random_seq
from racon_amd.layout import PairSet, ReadSet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from racon_amd import polisher
    polisher.build()
    return polisher


def _run(pairs_qt, strands=None):
    """pairs_qt: [(query segment as the aligner sees it on the FORWARD read, target segment)] -> (cigars, distances)."""
    from racon_amd.engine import HipEngine
    seqs, plist = [], []
    for k, (q, t) in enumerate(pairs_qt):
        seqs.append((t, None))
    nt = len(seqs)
    for k, (q, t) in enumerate(pairs_qt):
        seqs.append((q, None))
        plist.append((nt + k, k, 0 if strands is None else strands[k], 0, len(q), 0, len(t)))
    reads = ReadSet.from_sequences(seqs, nt)
    eng = HipEngine(3, -5, -4, True)
    eng.align_pairs(reads, PairSet.from_lists(plist))
    cig, dist = eng.alignment_cigars()
    return cig, dist, eng.align_stats()


def test_small_pairs_against_oracle_and_host(P):
    from oracle import nw_oracle
    rng = np.random.default_rng(9100)
    pairs = [(b"A", b"A"), (b"A", b"C"), (b"ACGT", b"A"), (b"A", b"ACGT"), (b"ACGTACGT", b"ACGTACGT"), (b"AAAA", b"TTTT"),
             (b"ACGTNNACGT", b"ACGTACGT"), (b"GATTACA" * 30, b"GATACA" * 30)]
    pairs += [(b"", b"ACGT"), (b"ACGTT", b"")]                       # an empty side: all deletions / all insertions
    for n in [2, 3, 63, 64, 65, 127, 128, 129, 200, 500, 900, 1500]:
        t = random_seq(rng, n)
        pairs.append((mutate(rng, t, float(rng.choice([0.02, 0.1, 0.3]))), t))
    cig, dist, _ = _run(pairs)
    for (q, t), c, d in zip(pairs, cig, dist):
        ref, rd = nw_oracle.cigar(q, t)
        assert c == ref, (len(q), len(t))
        assert c == P.align_cigar(q, t).encode()
        assert d == rd


def test_reverse_strand_pairs(P):
    """strand = 1: the rows are the reverse complement of the stored query segment (reference src/overlap.cpp:193-195)."""
    from oracle import nw_oracle
    rng = np.random.default_rng(9101)
    pairs, strands = [], []
    for n in [50, 300, 1200, 2600]:
        t = random_seq(rng, n, b"ACGTN")
        q_on_strand = mutate(rng, t, 0.1, b"ACGTN")
        pairs.append((nw_oracle.reverse_complement(q_on_strand), t)); strands.append(1)
        pairs.append((q_on_strand, t)); strands.append(0)
    cig, dist, _ = _run(pairs, strands)
    for k, ((q, t), c) in enumerate(zip(pairs, cig)):
        qs = nw_oracle.reverse_complement(q) if strands[k] else q
        assert c == P.align_cigar(qs, t).encode(), (k, len(q))


@pytest.mark.parametrize("seed,n,rate,indels", [(1, 2300, 0.12, 0), (2, 5200, 0.1, 2), (3, 9800, 0.13, 3), (4, 14000, 0.08, 4)])
def test_hirschberg_sized_pairs(P, seed, n, rate, indels):
    """Above ~1830 x 1830 the reference rule splits on the target axis (smallest optimal query split): one to three
    levels of recursion here, several 64-word passes per column sweep for the longer ones."""
    from oracle import nw_oracle
    rng = np.random.default_rng(9200 + seed)
    t = random_seq(rng, n)
    q = mutate(rng, t, rate, long_indels=indels)
    cig, dist, st = _run([(q, t), (t, q)])
    assert cig[0] == P.align_cigar(q, t).encode()
    assert cig[1] == P.align_cigar(t, q).encode()
    assert dist[0] == dist[1] == P.edit_distance(q, t)
    if n <= 5200:
        assert cig[0] == nw_oracle.cigar(q, t)[0]
    assert st["cells"] == 2 * len(q) * len(t)


def test_two_plane_pairs_and_symbols_only_the_target_has(P):
    """Four query symbols or fewer and no target symbol outside them: two bit planes (k_pair_align); a target symbol the query
    lacks (an N in the contig, a T where the read has none) must never match and sends the pair to three planes.  Both strands,
    sizes with several levels of splits (inherited column vectors, two sub-problems at a time)."""
    from oracle import nw_oracle
    rng = np.random.default_rng(9250)
    pairs, strands = [], []
    for n in [700, 3100, 6400]:
        t = random_seq(rng, n)                                            # ACGT
        q = mutate(rng, t, 0.1)
        pairs.append((q, t)); strands.append(0)
        pairs.append((nw_oracle.reverse_complement(q), t)); strands.append(1)
        tn = bytearray(t)
        for k in rng.integers(0, n, n // 50):
            tn[int(k)] = ord("N")                                         # only the target holds N
        pairs.append((q, bytes(tn))); strands.append(0)
        pairs.append((nw_oracle.reverse_complement(q), bytes(tn))); strands.append(1)
    t3 = random_seq(rng, 2800, b"ACG")
    pairs.append((mutate(rng, t3, 0.08, b"ACG"), random_seq(rng, 2700))); strands.append(0)          # three query symbols, T only in the target
    pairs.append((mutate(rng, t3, 0.08, b"ACG"), t3)); strands.append(1)
    t2 = random_seq(rng, 2600, b"AT")
    pairs.append((nw_oracle.reverse_complement(mutate(rng, t2, 0.1, b"AT")), t2)); strands.append(1)   # two symbols that complement into each other
    cig, dist, _ = _run(pairs, strands)
    for k, ((q, t), c, d) in enumerate(zip(pairs, cig, dist)):
        qs = nw_oracle.reverse_complement(q) if strands[k] else q
        assert c == P.align_cigar(qs, t).encode(), (k, len(q), len(t))
        assert d == P.edit_distance(qs, t)


def test_leaf_pairs_of_uneven_shapes(P):
    """Two leaves of one overlap run side by side in one wave when their words fit it together (pair_leaf_two: ACGT reads, two symbol
    planes): rectangles of every aspect -- leaves of unequal word counts, leaves that end at different columns, a leaf next to a split,
    both strands -- against the host aligner, CIGAR for CIGAR."""
    from oracle import nw_oracle
    rng = np.random.default_rng(9260)
    pairs, strands = [], []
    for (m, n) in [(3000, 9000), (9000, 3000), (4100, 4100), (2500, 7000), (7000, 2500), (5000, 3500), (3700, 5200), (12000, 2000), (2000, 12000),
                   (6400, 6300), (8200, 4000), (1900, 3900), (3900, 1900)]:
        t = random_seq(rng, n)
        q = mutate(rng, (t * (m // n + 1))[:m] if m > n else t[:m], 0.1)
        if m > n:                                     # a query longer than the target: the tail is unrelated sequence (long runs of I)
            q = q[:n] + random_seq(rng, m - n)
        k = len(pairs)
        pairs.append((q, t)); strands.append(0)
        if k % 2 == 0:
            pairs.append((nw_oracle.reverse_complement(q), t)); strands.append(1)
    cig, dist, _ = _run(pairs, strands)
    for k, ((q, t), c, d) in enumerate(zip(pairs, cig, dist)):
        qs = nw_oracle.reverse_complement(q) if strands[k] else q
        assert c == P.align_cigar(qs, t).encode(), (k, len(q), len(t))
        assert d == P.edit_distance(qs, t)


def test_degenerate_shapes_and_wide_alphabets(P):
    rng = np.random.default_rng(9300)
    iupac = b"ACGTNRYKMSWBDHV"
    pairs = []
    t = random_seq(rng, 9000)
    pairs.append((t[100:140], t))                       # 40 rows x 9000 columns: one word, a leaf far above 1830 columns
    pairs.append((t, t[4000:4040]))                     # 9000 rows x 40 columns: 141 words in three passes, still a leaf
    pairs.append((t, t[4000:4003]))
    pairs.append((t[:5000], random_seq(rng, 4800)))     # unrelated sequences: distance near the maximum
    pairs.append((random_seq(rng, 3000, b"A"), random_seq(rng, 2900, b"A")))     # one symbol: every path co-optimal
    u = random_seq(rng, 2500, iupac)
    pairs.append((mutate(rng, u, 0.1, iupac), u))       # fifteen symbols: the 8-plane path
    low = bytes(rng.integers(1, 256, 1500, dtype=np.int64).astype(np.uint8).tolist())
    pairs.append((mutate(rng, low, 0.05, low[:50]), low))   # arbitrary bytes
    cig, dist, _ = _run(pairs)
    for (q, tt), c, d in zip(pairs, cig, dist):
        assert c == P.align_cigar(q, tt).encode(), (len(q), len(tt))
        assert d == P.edit_distance(q, tt)


def _pairs_from_alignments(reads, al):
    """PairSet of the overlaps of a CigarSet: the query segment on the forward read from q_start / strand / CIGAR."""
    off = reads.seq_off
    rows = []
    for o in range(al.n_overlaps):
        c = al.cigar[int(al.cigar_off[o]):int(al.cigar_off[o + 1])].tobytes()
        qspan, num = 0, 0
        for ch in c:
            if 48 <= ch <= 57:
                num = num * 10 + ch - 48
            else:
                if ch in b"MI=X":
                    qspan += num
                num = 0
        ql = int(off[int(al.q_id[o]) + 1] - off[int(al.q_id[o])])
        if al.strand[o]:
            q_end = ql - int(al.q_start[o]); q_begin = q_end - qspan
        else:
            q_begin = int(al.q_start[o]); q_end = q_begin + qspan
        rows.append((int(al.q_id[o]), int(al.t_id[o]), int(al.strand[o]), q_begin, q_end, int(al.t_begin[o]), int(al.t_end[o])))
    return PairSet.from_lists(rows)


@pytest.mark.parametrize("ovl,reads_file", [("sample_overlaps.paf.gz", "sample_reads.fastq.gz"), ("sample_ava_overlaps.paf.gz", "sample_reads.fastq.gz")])
def test_reference_paf_overlaps(P, ovl, reads_file):
    """Every overlap of the reference's PAF samples (contig polishing and the all-vs-all fragment set): device CIGAR ==
    host CIGAR (the host's reproduce goldens 1312 / 1566 / 40-401246 end to end), then the windows built from the device
    alignments equal the windows built from the host's CIGARs, array for array."""
    from racon_amd.engine import HipEngine
    target = "sample_layout.fasta.gz" if "ava" not in ovl else reads_file
    typ = "kC" if "ava" not in ovl else "kF"
    p = P.Polisher(REFDATA + reads_file, REFDATA + ovl, REFDATA + target, typ, 500, 10.0, 0.3, True, 5, -4, -8, num_threads=8)
    p.initialize(keep_layout=True)
    reads, _, wt, wl, qt = p.layout()
    al = p.alignments()
    pairs = _pairs_from_alignments(reads, al)
    eng = HipEngine(5, -4, -8, True)
    eng.align_pairs(reads, pairs)
    cig, dist = eng.alignment_cigars()
    host = [al.cigar[int(al.cigar_off[o]):int(al.cigar_off[o + 1])].tobytes() for o in range(al.n_overlaps)]
    bad = [o for o in range(al.n_overlaps) if cig[o] != host[o]]
    assert not bad, (len(bad), bad[:5])
    eng.build_windows_from_cigars(reads, al, wl, qt, wt)
    want = eng.export_batch()
    eng2 = HipEngine(5, -4, -8, True)
    eng2.build_windows_from_pairs(reads, pairs, wl, qt, wt)
    got = eng2.export_batch()
    for name in ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    r1, r2 = eng.run(), eng2.run()
    assert r1.consensus == r2.consensus


@pytest.mark.parametrize("ovl", ["paf"])
def test_cli_with_device_alignment(P, oracle, ovl, tmp_path_factory):
    """RACON_HIP_DEVICE_WINDOWS=3: the host parses; alignment, breaking points, window construction and consensus all
    run on the device -- same FASTA as host layer + oracle, byte for byte."""
    import subprocess
    from racon_amd.synth import simulate_files
    d = str(tmp_path_factory.mktemp("e2e3"))
    paths, _ = simulate_files(d, contig_len=20000, coverage=25.0, read_len=3000, n_contigs=2)
    p = P.Polisher(paths["reads"], paths[ovl], paths["targets"], "kC", 500, 10.0, 0.3, True, 3, -5, -4, num_threads=4)
    p.initialize()
    b = p.windows()
    ref = p.assemble(oracle.consensus(b, 3, -5, -4, True, 0), True)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "racon_amd", "host", "racon_hip")
    env = dict(os.environ, RACON_HIP_DEVICE_WINDOWS="3")
    out = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout
    assert out == ref


def test_cli_device_alignment_falls_back_to_the_host_aligner(P, oracle, tmp_path_factory):
    """An overlap set the device aligner has no room for (RCN_E_CAPACITY / RCN_E_NOMEM) is aligned by the host's
    edlib-equivalent instead and goes on through the device's CIGAR path: same FASTA.  RACON_HIP_FORCE_ALIGN_FALLBACK takes
    that road for every shard."""
    import subprocess
    from racon_amd.synth import simulate_files
    d = str(tmp_path_factory.mktemp("e2e4"))
    paths, _ = simulate_files(d, contig_len=15000, coverage=20.0, read_len=3000, n_contigs=2)
    p = P.Polisher(paths["reads"], paths["paf"], paths["targets"], "kC", 500, 10.0, 0.3, True, 3, -5, -4, num_threads=4)
    p.initialize()
    ref = p.assemble(oracle.consensus(p.windows(), 3, -5, -4, True, 0), True)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "racon_amd", "host", "racon_hip")
    for shards in ("1", "3"):
        env = dict(os.environ, RACON_HIP_DEVICE_WINDOWS="3", RACON_HIP_FORCE_ALIGN_FALLBACK="1", RACON_HIP_DEVICE_SHARDS=shards)
        out = subprocess.run([exe, "-t", "4", paths["reads"], paths["paf"], paths["targets"]], check=True, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout
        assert out == ref, shards
