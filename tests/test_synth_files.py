"""The file form of the synthetic workloads (racon_amd.synth.simulate_window_files) against their packed form
(simulate_windows_parallel): the windows racon's initialize() cuts from the files -- host layer, reference
src/polisher.cpp:388-461 + src/overlap.cpp:226-292 -- are the packed batch, array for array.  bench.py times the
product (files -> Polisher::polish) and the kernel (packed batch) on ONE workload because of this."""
import numpy as np

from racon_amd.polisher import Polisher, build
from racon_amd.synth import simulate_window_files, simulate_windows_parallel

FIELDS = ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals")


def test_files_give_the_packed_windows(tmp_path):
    build()
    for contig, piece, seed in ((60000, 25000, 7), (20000, 1_000_000, 20260921)):
        paths = simulate_window_files(str(tmp_path / ("d%d" % seed)), contig, 30.0, 10000, seed=seed, piece=piece, workers=3)
        ref = simulate_windows_parallel(contig, 500, 30.0, 10000, seed=seed, piece=piece, workers=3)
        p = Polisher(paths["reads"], paths["sam"], paths["targets"], num_threads=4)
        p.initialize()
        assert p.num_windows() == ref.n_windows
        got = p.windows()
        for f in FIELDS:
            assert np.array_equal(getattr(got, f), getattr(ref, f)), f
        p.close()


def test_paf_overlaps_resolve_to_the_same_layers(tmp_path):
    """The PAF twin of the file set names the same overlaps (pre-aligned on the host, so begin/end may move by a base)."""
    build()
    paths = simulate_window_files(str(tmp_path), 20000, 20.0, 5000, seed=11, workers=1)
    a = Polisher(paths["reads"], paths["sam"], paths["targets"], num_threads=2)
    b = Polisher(paths["reads"], paths["paf"], paths["targets"], num_threads=2)
    a.initialize(); b.initialize()
    wa, wb = a.windows(), b.windows()
    assert wa.n_windows == wb.n_windows and abs(wa.n_seqs - wb.n_seqs) <= wa.n_windows
    a.close(); b.close()
