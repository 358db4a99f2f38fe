"""End to end through the drop-in surface: racon's command line (`racon_hip`, reference src/main.cpp) and
`Polisher::{initialize, polish}` on a small synthetic data set written in racon's own input formats.

not gpu: host layer + oracle backend — the polished contig must be much closer to the truth than the draft
         (the pipeline really polishes), SAM and PAF inputs agree on the window count.
gpu    : the `racon_hip` binary and `Polisher.polish()` (both run the consensus stage on the MI355X through
         libracon_hip.so) print byte-identical FASTA to host layer + oracle.
"""
import os
import subprocess

import pytest

from racon_amd.synth import simulate_files

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def P():
    from racon_amd import polisher
    polisher.build()
    return polisher


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("e2e"))
    return simulate_files(d, contig_len=20000, coverage=25.0, read_len=3000, n_contigs=2)


def _oracle_fasta(P, oracle, paths, ovl, scores=(3, -5, -4), **kw):
    p = P.Polisher(paths["reads"], paths[ovl], paths["targets"], "kC", 500, 10.0, 0.3, True, *scores, num_threads=4, **kw)
    p.initialize()
    b = p.windows()
    return p.assemble(oracle.consensus(b, *scores, True, 0), True), b.n_windows


def test_pipeline_polishes(P, oracle, data):
    paths, truth = data
    fa_sam, n_sam = _oracle_fasta(P, oracle, paths, "sam")
    fa_paf, n_paf = _oracle_fasta(P, oracle, paths, "paf")
    assert n_sam == n_paf == 2 * 40
    import gzip
    draft = [l for l in gzip.open(paths["targets"]).read().split(b"\n") if l and not l.startswith(b">")]
    for fa in (fa_sam, fa_paf):
        seqs = P.parse_fasta(fa)
        assert len(seqs) == 2
        for (hdr, s), t, d in zip(seqs, truth, draft):
            assert b"LN:i:" in hdr and b"RC:i:" in hdr and b"XC:f:" in hdr           # reference src/polisher.cpp:521-526
            assert oracle.edit_distance(s, t) * 5 < oracle.edit_distance(d, t)        # 3 % draft errors -> well under 0.6 %


@pytest.mark.gpu
@pytest.mark.parametrize("ovl", ["sam", "paf"])
def test_cli_matches_oracle(P, oracle, data, ovl):
    """The binary as a user runs it (windows built in HBM at the end of initialize(): its default when the job fits the device) and
    with RACON_HIP_DEVICE_WINDOWS=0 (Window::add_layer on the host, chunks packed inside polish()): the FASTA of host layer + oracle."""
    paths, _ = data
    ref, _ = _oracle_fasta(P, oracle, paths, ovl)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    env = {k: v for k, v in os.environ.items() if k != "RACON_HIP_DEVICE_WINDOWS"}
    run = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.stdout == ref and b"transformed data into windows (on the device)" in run.stderr
    env["RACON_HIP_DEVICE_WINDOWS"] = "0"
    host = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert host.stdout == ref and b"(on the device)" not in host.stderr


@pytest.mark.gpu
def test_polisher_polish_and_options(P, oracle, data):
    paths, _ = data
    # test-suite scores, larger window, unpolished targets kept, no trimming
    for scores, w, trim in [((5, -4, -8), 500, True), ((1, -1, -1), 1000, False)]:
        p = P.Polisher(paths["reads"], paths["sam"], paths["targets"], "kC", w, 10.0, 0.3, trim, *scores, num_threads=4)
        p.initialize()
        b = p.windows()
        ref = p.assemble(oracle.consensus(b, *scores, trim, 0), False)
        p2 = P.Polisher(paths["reads"], paths["sam"], paths["targets"], "kC", w, 10.0, 0.3, trim, *scores, num_threads=4)
        p2.initialize()
        assert p2.polish(False) == ref


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("ovl", ["sam", "paf"])
def test_cli_with_device_side_windows_matches_oracle(P, oracle, data, ovl, mode):
    """RACON_HIP_DEVICE_WINDOWS=1: the host only parses and finds breaking points; the windows are cut, filtered and packed
    in HBM (rcn_engine_build_windows, reference src/polisher.cpp:388-461) and polished there — same FASTA, byte for byte."""
    paths, _ = data
    ref, _ = _oracle_fasta(P, oracle, paths, ovl)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    # mode 2: the CIGAR walk (breaking points, reference src/overlap.cpp:226-292) runs on the device as well
    env = dict(os.environ, RACON_HIP_DEVICE_WINDOWS=mode)
    run = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.stdout == ref
    # the windows are built where the reference builds them -- at the end of initialize() -- and stay resident: polish() is the
    # consensus alone; RACON_HIP_BUILD_IN_POLISH=1 (construction inside polish(), shard after shard) prints the same FASTA
    assert b"transformed data into windows (on the device)" in run.stderr
    env["RACON_HIP_BUILD_IN_POLISH"] = "1"
    late = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert late.stdout == ref and b"(on the device)" not in late.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ovl,mode", [("sam", "1"), ("sam", "2"), ("paf", "2"), ("paf", "3")])
def test_cli_device_windows_sharded(P, oracle, data, ovl, mode):
    """RACON_HIP_DEVICE_SHARDS=3: the device-side construction cut into three window ranges (what the host layer does with
    one range per device on a multi-GPU node; here the three shards share the one device): overlaps across a boundary go
    to both sides, every window is taken from the shard that owns it -- same FASTA."""
    paths, _ = data
    ref, _ = _oracle_fasta(P, oracle, paths, ovl)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    env = dict(os.environ, RACON_HIP_DEVICE_WINDOWS=mode, RACON_HIP_DEVICE_SHARDS="3")
    out = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout
    assert out == ref


@pytest.mark.gpu
@pytest.mark.parametrize("ovl,mode", [("sam", "0"), ("paf", "0"), ("sam", "2"), ("paf", "3")])
def test_cli_on_three_logical_devices(P, oracle, data, ovl, mode):
    """RACON_HIP_FAKE_DEVICES=3: the host layer drives three logical devices (all mapped onto the one GPU of the box) --
    six engines pulling small chunks from the shared cursor (mode 0, reference src/cuda/cudapolisher.cpp:254-276,336-350),
    or one window range per device with only that range's reads uploaded (device-built windows) -- same FASTA."""
    paths, _ = data
    ref, _ = _oracle_fasta(P, oracle, paths, ovl)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    env = dict(os.environ, RACON_HIP_FAKE_DEVICES="3", RACON_HIP_CHUNK_WINDOWS="7", RACON_HIP_DEVICE_WINDOWS=mode)
    env["RACON_HIP_TIMING"] = "1"
    run = subprocess.run([exe, "-t", "4", paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.stdout == ref
    if mode == "0":
        # RACON_HIP_CHUNK_WINDOWS is an exact window count (no byte floor): many chunks, and more than one engine took some
        import re
        took = re.findall(rb"timing: engine (\d+) chunk (\d+) \((\d+) windows\)", run.stderr)
        assert len(took) >= 3 and all(int(n) <= 7 for _, _, n in took), run.stderr[-2000:]
        assert len({e for e, _, _ in took}) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("flags,ovl,aligner_on_device", [
    (["--cudaaligner-batches", "1"], "paf", True),        # reference src/main.cpp:125-127 -> src/polisher.cpp:137-147
    (["--cudaaligner-batches", "2"], "sam", False),       # CIGARs in the file: nothing to align, the device walks them
    (["--cudaaligner-batches", "0"], "paf", False),       # 0 = the reference's default: host pre-alignment
    (["-b"], "paf", False),                               # banded approximation: accepted, ignored (exact DP)
    (["--cudaaligner-band-width", "128", "--cudaaligner-batches", "1"], "paf", True),
    (["-c", "2", "--cudaaligner-batches", "1", "-b"], "paf", True),
])
def test_cli_reference_cuda_flags(P, oracle, data, flags, ovl, aligner_on_device):
    """The reference's CUDA options keep the reference's meaning (src/main.cpp:117-131): `--cudaaligner-batches n > 0` moves
    the overlap alignment to the device (the byte-exact pair aligner + window construction in HBM), `-c` is engines per
    device, `-b` / `--cudaaligner-band-width` select approximations that do not exist here.  FASTA identical in every case."""
    paths, _ = data
    ref, _ = _oracle_fasta(P, oracle, paths, ovl)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    env = {k: v for k, v in os.environ.items() if k != "RACON_HIP_DEVICE_WINDOWS"}
    run = subprocess.run([exe, "-t", "4"] + flags + [paths["reads"], paths[ovl], paths["targets"]], check=True, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.stdout == ref
    assert (b"left the overlaps to the device aligner" in run.stderr) == aligner_on_device, run.stderr[-1500:]


def test_cli_fails_loudly_without_a_device(data):
    """No MI355X (or no libracon_hip.so): the product does not fall back to anything -- `racon_hip` parses its input, then exits non-zero
    with the reason, and prints no FASTA.  (Runs where there is no GPU: the CPU test tier; skipped on a GPU box.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    paths, _ = data
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    for env_add in ({}, {"RACON_HIP_DEVICE_WINDOWS": "0"}, {"RACON_HIP_DEVICE_WINDOWS": "2"}):
        run = subprocess.run([exe, "-t", "2", paths["reads"], paths["sam"], paths["targets"]], env=dict(os.environ, **env_add),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert run.returncode != 0 and run.stdout == b"", env_add
        assert b"no MI355X device / libracon_hip.so available (the consensus stage has no CPU fallback)" in run.stderr, run.stderr[-500:]


def test_cli_help_names_the_aligner_flag():
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    out = subprocess.run([exe, "--help"], check=True, stdout=subprocess.PIPE).stdout
    assert b"--cudaaligner-batches <int>" in out and b"--cudaaligner-band-width" in out and b"-c, --cudapoa-batches" in out
    # the flags parse (and the factory errors come after them, reference test/racon_test.cpp:60-84)
    run = subprocess.run([exe, "--cudaaligner-batches", "1", "-b", "--cudaaligner-band-width", "64", "a.txt", "b.paf", "c.fasta"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert run.returncode != 0 and b"unsupported format extension" in run.stderr


@pytest.mark.gpu
def test_cli_survives_a_device_short_of_memory(tmp_path):
    """A chunk the device has no room for is polished in halves, on the GPU (racon_amd/host/polisher.cpp; the reference
    completes its run when a batch fails: src/cuda/cudapolisher.cpp:357-373).  RCN_FAIL_ALLOC_ABOVE (a test switch behind
    RCN_EXPERIMENT=1) makes every device allocation above 1 GB fail like an exhausted device: 1000 windows want ~8 GB of
    scratch at once, the run completes in pieces of 125 with the FASTA of the unrestricted run."""
    from racon_amd.synth import simulate_window_files
    paths = simulate_window_files(str(tmp_path), 500_000, 30.0, 10000, seed=20260931, workers=4)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    cmd = [exe, "-t", "8", paths["reads"], paths["sam"], paths["targets"]]
    free = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    # (the host-built path: chunks streamed through the engines, what a job too large for the device-side construction takes)
    env = dict(os.environ, RCN_EXPERIMENT="1", RCN_FAIL_ALLOC_ABOVE=str(1 << 30), RACON_HIP_DEVICE_WINDOWS="0")
    tight = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert tight.stdout == free.stdout and tight.stdout.count(b">") == 1
    assert b"polishing them in halves" in tight.stderr and b"polishing them in halves" not in free.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_cli_short_reads_through_the_small_window_kernel(P, oracle, tmp_path_factory, mode):
    """racon's short-read use (kNGS windows: reference src/polisher.cpp:229-230 picks the type from the mean read length) with
    `-w 200`: 150-base reads at 60x on two contigs.  The windows have the small-window kernel's shape
    (racon_amd/csrc/poa_small.hpp: one wave per window, graph in LDS) whether the host builds them (mode 0) or the device does
    (1: windows, 2: CIGAR walk too) -- the FASTA of host layer + oracle, byte for byte."""
    d = str(tmp_path_factory.mktemp("short"))
    paths, _ = simulate_files(d, contig_len=30000, coverage=60.0, read_len=150, n_contigs=2, sub=0.004, ins=0.0005, dele=0.0005, backbone_errors=0.01)
    p = P.Polisher(paths["reads"], paths["sam"], paths["targets"], "kC", 200, 10.0, 0.3, True, 3, -5, -4, num_threads=4)
    p.initialize()
    b = p.windows()
    assert b.n_windows == 2 * 150 and int(b.win_type.max()) == 0          # kNGS
    ref = p.assemble(oracle.consensus(b, 3, -5, -4, True, 0), True)
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    env = dict(os.environ, RCN_DEBUG="1")
    if mode != "0":
        env["RACON_HIP_DEVICE_WINDOWS"] = mode
    r = subprocess.run([exe, "-t", "4", "-w", "200", paths["reads"], paths["sam"], paths["targets"]], check=True, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.stdout == ref
    assert b"small-window kernel" in r.stderr                               # (the engine's debug line of the pass)
