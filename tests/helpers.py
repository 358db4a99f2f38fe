"""Shared test helpers: hand-made edge-case windows and comparison utilities."""
import os

import numpy as np

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_windows


# The reference's own test inputs (reference test/data/*, 3 MB of gz files), committed as fixtures so that the
# end-to-end goldens of reference test/racon_test.cpp:86-295 also run on the GPU box, where /root/reference does
# not exist.  tests/test_reference_goldens.py::test_refdata_is_the_reference_data checks them byte for byte
# against /root/reference wherever that is present.
REFDATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdata") + os.sep


def q(s, v=20):
    return bytes([33 + v]) * len(s)


def edge_case_windows():
    """Windows exercising the corners the reference's Window API allows
    (reference src/window.cpp:42-63, 65-71, 88-107, 125-146)."""
    bb = b"ACGTACGTTAGCTAGCTAGGATCCATGCATGCAAATTTCCCGGGATATCGCGTTAACCGGTTAACGTAGCTAGCATCGATCGGCTAGCTAACGT"
    L = len(bb)
    wins = []
    # 0: backbone only  (<3 sequences -> backbone, not polished)
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0)]})
    # 1: backbone + one layer (still < 3)
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (bb, q(bb), 0, L - 1)]})
    # 2: two identical full-span layers
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (bb, q(bb), 0, L - 1), (bb, q(bb), 0, L - 1)]})
    # 3: layers without quality (weight 1), one with a substitution and an insertion
    l1 = bb[:30] + b"T" + bb[31:]
    l2 = bb[:50] + b"GG" + bb[50:]
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (l1, None, 0, L - 1), (l2, None, 0, L - 1), (l1, None, 0, L - 1)]})
    # 4: partial layers (Subgraph branch), ties in `begin` (unstable std::sort order matters)
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (bb[10:60], q(bb[10:60]), 10, 59), (bb[10:70], q(bb[10:70], 9), 10, 69),
                                     (bb[40:], q(bb[40:]), 40, L - 1), (bb[:35], q(bb[:35], 30), 0, 34), (bb[5:80], None, 5, 79)]})
    # 5: kNGS window: trimming must NOT apply; low coverage ends
    wins.append({"type": 0, "seqs": [(bb, q(bb, 0), 0, 0), (bb[20:70], q(bb[20:70]), 20, 69), (bb[22:72], q(bb[22:72]), 22, 71)]})
    # 6: same as 5 but kTGS -> trimmed to the covered part
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (bb[20:70], q(bb[20:70]), 20, 69), (bb[22:72], q(bb[22:72]), 22, 71),
                                     (bb[21:71], q(bb[21:71]), 21, 70), (bb[20:72], q(bb[20:72]), 20, 71)]})
    # 7: non-ACGT symbols (N and IUPAC), lower quality chars, real backbone qualities
    l3 = bb[:15] + b"N" + bb[16:40] + b"RY" + bb[42:]
    wins.append({"type": 1, "seqs": [(bb, q(bb, 7), 0, 0), (l3, q(l3, 3), 0, L - 1), (l3, q(l3, 40), 0, L - 1), (bb, q(bb, 12), 0, L - 1)]})
    # 8: one-base and two-base layers (no edge / single edge), plus a long overhanging layer
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (b"A", q(b"A"), 3, 4), (b"CG", q(b"CG"), 1, 3), (bb + b"ACGTACGT", q(bb + b"ACGTACGT"), 0, L - 1),
                                     (b"TTTT" + bb, q(b"TTTT" + bb), 0, L - 1)]})
    # 9: chimeric warning: >= 3 sequences but no position reaches the coverage threshold
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0)] + [(bb[i * 9:i * 9 + 9], q(bb[i * 9:i * 9 + 9]), i * 9, i * 9 + 8) for i in range(10)]})
    # 10: a layer completely different from the backbone
    junk = b"TTTTTTTTTTGGGGGGGGGGTTTTTTTTTTGGGGGGGGGGTTTTTTTTTT"
    wins.append({"type": 1, "seqs": [(bb, q(bb, 0), 0, 0), (junk, q(junk), 0, L - 1), (junk, q(junk), 0, L - 1), (bb, q(bb), 0, L - 1)]})
    # 11: tiny backbone
    wins.append({"type": 1, "seqs": [(b"ACG", q(b"ACG", 0), 0, 0), (b"ACG", q(b"ACG"), 0, 2), (b"AG", q(b"AG"), 0, 2), (b"ACCG", None, 0, 2)]})
    return wins


def edge_case_batch() -> WindowBatch:
    return WindowBatch.from_windows(edge_case_windows())


def synthetic_sets():
    """(name, batch, (m, x, g)) small seeded sets in the shapes of BASELINE.json's configs."""
    short = dict(sub=0.003, ins=0.0005, dele=0.0005, phred_mean=30, phred_sd=0)
    return [
        ("ont_w500_default", simulate_windows(15000, 500, 30, 4000, seed=101), (3, -5, -4)),
        ("ont_w500_test_scores", simulate_windows(10000, 500, 25, 4000, seed=102), (5, -4, -8)),
        ("ont_w500_edit", simulate_windows(8000, 500, 20, 4000, seed=103), (1, -1, -1)),
        ("ont_noqual", simulate_windows(8000, 500, 20, 4000, seed=104, with_quality=False), (3, -5, -4)),
        ("short_w200", simulate_windows(8000, 200, 60, 150, seed=105, **short), (3, -5, -4)),
        ("ont_w1000", simulate_windows(8000, 1000, 20, 5000, seed=106), (5, -4, -8)),
        ("noisy_backbone", simulate_windows(8000, 500, 20, 4000, seed=107, backbone_errors=0.05), (3, -5, -4)),
    ]


def assert_same(got, ref, tag=""):
    assert len(got.consensus) == len(ref.consensus)
    bad = [i for i in range(len(ref.consensus)) if got.consensus[i] != ref.consensus[i]]
    assert not bad, f"{tag}: {len(bad)} windows differ, first {bad[:5]}"
    assert (np.asarray(got.polished) == np.asarray(ref.polished)).all(), f"{tag}: polished flags differ"
    assert (np.asarray(got.chimeric) == np.asarray(ref.chimeric)).all(), f"{tag}: chimeric flags differ"
