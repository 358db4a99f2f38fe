"""Which layer of window 2 of f8913 goes wrong: the window cut after its first k sequences (backbone, then the layers by begin), k = 3 ... 24,
through the engine against the oracle."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from racon_amd.batch import WindowBatch
from racon_amd.polisher import Polisher
from racon_amd.engine import HipEngine
from oracle import oracle_lib
d = "tools/dbg_f8913"
p = Polisher(d + "/reads.fastq", d + "/overlaps.paf", d + "/target.fastq", "kF", 500, 10.0, 0.3, True, 3, -5, -4, 8, 1)
p.initialize(); b = p.windows()
W = b.window(2); seqs = W["seqs"]; ns = len(seqs)
order = [0] + sorted(range(1, ns), key=lambda i: (seqs[i][2], i))
e = HipEngine(3, -5, -4, True)
first = None
for k in range(3, ns + 1):
    w = WindowBatch.from_windows([{"type": W["type"], "seqs": [seqs[i] for i in order[:k]]}])
    o = oracle_lib.consensus(w, 3, -5, -4, True, 0, simd=True)
    g = e.consensus(w)
    same = g.consensus[0] == o.consensus[0]
    s = seqs[order[k - 1]]
    print("first %2d sequences: %s (last layer len %d, begin %d, end %d)" % (k, "same" if same else "DIFFER", len(s[0]), s[2], s[3]))
    if not same and first is None: first = k
print("first failing k:", first)
