"""-m gpu: window shapes the reference's CPU path takes without a cap (reference src/window.cpp:42-63,65-149 have no
depth or length limit; its CUDA path caps at 200 layers / 1023 bases, src/cuda/cudapolisher.cpp:226,
src/cuda/cudabatch.cpp:56-59 -- the caps this engine promises not to have): hundreds of layers on a 500-bp window
(repeat / plasmid pile-ups), windows of 3000 and 5000 bp (layers beyond the 2048 columns of the int16 kernel's widest
row shape: the int32 kernel is the only path), lower-case and N-run reads through the command line."""
import os
import subprocess

import numpy as np
import pytest

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_window_files, simulate_windows
from helpers import assert_same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("depth,scores", [(300, (3, -5, -4)), (300, (5, -4, -8)), (1000, (3, -5, -4))])
def test_hundreds_of_layers_on_a_500_bp_window(oracle, depth, scores):
    from racon_amd.engine import HipEngine
    b = simulate_windows(1500, 500, 2.0 * depth, 3000, seed=4000 + depth)      # `depth` reads, two of three windows under all of them
    layers = np.diff(b.win_seq_off.astype(np.int64)) - 1
    assert b.n_windows == 3 and layers.max() >= depth * 0.8
    # next to ordinary windows in one batch: per-window output offsets and capacities differ by two orders of magnitude
    both = b.concat(simulate_windows(20000, 500, 30.0, 4000, seed=4001))
    eng = HipEngine(*scores, True)
    got = eng.consensus(both)
    assert_same(got, oracle.consensus(both, *scores, True, 0, simd=True), "%d layers" % depth)
    eng.upload(both)
    assert_same(eng.run(), got, "%d layers, resident" % depth)


@pytest.mark.parametrize("w,scores", [(3000, (3, -5, -4)), (5000, (3, -5, -4)), (3000, (1, -1, -1))])
def test_windows_longer_than_the_int16_row_shapes(oracle, w, scores):
    from racon_amd.engine import HipEngine
    b = simulate_windows(4 * w + w // 3, w, 20.0, 3 * w, seed=5000 + w)
    lens = np.diff(b.seq_off.astype(np.int64))
    assert b.n_windows == 5 and lens.max() > 2048
    eng = HipEngine(*scores, True)
    got = eng.consensus(b)
    assert_same(got, oracle.consensus(b, *scores, True, 0), "-w %d" % w)
    assert eng.stats()["n_retried"] >= 4                      # (the int32 kernel took them)


def test_lower_case_and_n_runs_through_the_cli(oracle, tmp_path):
    """Sequence upper-cases its data (reference src/sequence.cpp:24-27); N (any byte) is a symbol like the others."""
    from racon_amd import polisher as P
    P.build()
    paths = simulate_window_files(str(tmp_path), 40000, 25.0, 6000, seed=31, workers=1)
    rng = np.random.default_rng(31)
    lines = open(paths["reads"], "rb").read().split(b"\n")
    for i in range(1, len(lines), 4):
        s = bytearray(lines[i])
        if (i // 4) % 2 == 0:
            s = bytearray(bytes(s).lower())
        for _ in range(3):                                    # runs of N inside the read (same length: the CIGARs stay valid)
            a = int(rng.integers(0, max(1, len(s) - 40))); n = int(rng.integers(1, 40))
            s[a:a + n] = (b"n" if (i // 4) % 2 == 0 else b"N") * len(s[a:a + n])
        lines[i] = bytes(s)
    open(paths["reads"], "wb").write(b"\n".join(lines))
    p = P.Polisher(paths["reads"], paths["sam"], paths["targets"], "kC", 500, 10.0, 0.3, True, 3, -5, -4, num_threads=4)
    p.initialize()
    b = p.windows()
    assert b"N" in b.bases.tobytes() and not any(c in b.bases.tobytes() for c in (b"a", b"c", b"g", b"t", b"n"))
    ref = p.assemble(oracle.consensus(b, 3, -5, -4, True, 0), True)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    for mode in ("0", "2"):
        env = dict(os.environ)
        if mode != "0":
            env["RACON_HIP_DEVICE_WINDOWS"] = mode
        out = subprocess.run([exe, "-t", "4", paths["reads"], paths["sam"], paths["targets"]], check=True, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout
        assert out == ref, mode
