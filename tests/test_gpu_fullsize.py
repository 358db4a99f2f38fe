"""-m gpu, BASELINE.json's cfg2 at full size (1 Mbp contig, 30x, w=500: 2000 windows, 24.5 G DP cells), through
properties that do not need the oracle at that size:
  * determinism / idempotence: two launches over the resident batch give identical bytes;
  * independence (what sharding over GPUs relies on, reference src/polisher.cpp:496-503): polishing a shard alone
    gives exactly the slice of the full result, for an uneven 3-way split;
  * a checksum of checksums over all windows equals the one of the sharded runs;
  * spot check: a seeded sample of windows against the oracle.
"""
import hashlib

import numpy as np
import pytest

from racon_amd.synth import config_windows

pytestmark = pytest.mark.gpu


def _digest(cons):
    h = hashlib.sha256()
    for c in cons:
        h.update(hashlib.md5(c).digest())
    return h.hexdigest()


def test_cfg2_full_size_properties(oracle):
    from racon_amd.engine import HipEngine
    b = config_windows("cfg2")
    assert b.n_windows == 2000
    eng = HipEngine(3, -5, -4, True)
    eng.upload(b)
    r1 = eng.run()
    r2 = eng.run()
    assert r1.consensus == r2.consensus and (r1.polished == r2.polished).all()
    st = eng.stats()
    assert st["n_retried"] == 0 and st["dp_cells_full"] > 2.0e10 and st["dp_cells"] <= st["dp_cells_full"]
    # shards: each rank's result is the slice of the full result
    parts = []
    for rank in range(3):
        sub, idx = b.shard(rank, 3)
        rs = HipEngine(3, -5, -4, True).consensus(sub)
        assert rs.consensus == [r1.consensus[int(i)] for i in idx]
        parts += rs.consensus
    assert _digest(parts) == _digest(r1.consensus)
    # consensus lengths stay near the window length (TGS trimming may shorten the ends)
    lens = np.array([len(c) for c in r1.consensus])
    assert 400 < np.median(lens) < 520
    # oracle on a seeded sample
    rng = np.random.default_rng(1)
    pick = sorted(rng.choice(b.n_windows, 48, replace=False).tolist())
    ref = oracle.consensus(b.select(pick), 3, -5, -4, True, 0)
    assert ref.consensus == [r1.consensus[i] for i in pick]


def _all_windows_vs_oracle(oracle, b, scores, tag, trim=True):
    """Every window of `b` against the oracle (its AVX2 int16 variant on all host threads: byte-identical to the scalar
    oracle, tests/test_oracle_spec.py), byte for byte, flags included."""
    import os
    from helpers import assert_same
    from racon_amd.engine import HipEngine
    eng = HipEngine(*scores, trim)
    got = eng.consensus(b)
    ref = oracle.consensus(b, *scores, trim, os.cpu_count() or 1, simd=True)
    assert_same(got, ref, tag)
    return eng.stats()


def test_cfg2_every_window_against_the_oracle(oracle):
    """BASELINE configs[1] at its stated size: all 2000 windows, not a sample."""
    b = config_windows("cfg2")
    assert b.n_windows == 2000
    st = _all_windows_vs_oracle(oracle, b, (3, -5, -4), "cfg2 full size")
    assert st["n_retried"] == 0


def test_cfg4_full_size_against_the_oracle(oracle):
    """BASELINE configs[3] at the size SURVEY 8(d) gives it: 1 Mbp, 150 bp reads at 60x, -w 200 -> 5000 kNGS windows of
    ~140 short fragments each, almost every layer through the Subgraph branch (reference src/window.cpp:99-107)."""
    b = config_windows("cfg4")
    assert b.n_windows == 5000 and int(b.win_type.max()) == 0
    _all_windows_vs_oracle(oracle, b, (3, -5, -4), "cfg4 full size")


def test_w1000_full_size_against_the_oracle(oracle):
    """-w 1000 on 1 Mbp (reference test/racon_test.cpp:179-197 uses w = 1000 on the sample): 1000 windows, layers of
    ~1000 bases (the 4-wave DP shapes), at the reference tests' scores where |g| (V + l) leaves the int16 range."""
    b = config_windows("w1000")
    assert b.n_windows == 1000
    _all_windows_vs_oracle(oracle, b, (5, -4, -8), "w1000 full size, scores 5/-4/-8")
    _all_windows_vs_oracle(oracle, b, (3, -5, -4), "w1000 full size, scores 3/-5/-4")


def test_cfg3_whole_job_on_one_gpu_against_the_oracle(oracle):
    """BASELINE configs[2] at its stated size on ONE device: the 50 Mbp / 100 000-window job (the windows bench.py polishes with
    `--config cfg3`), every window against the oracle -- one launch over a queue fifty times the resident slots.  (Eight devices
    polish an eighth of these windows each: windows are independent, reference src/polisher.cpp:496-503.)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    b = bench.cached_windows(50_000_000, 500, 30.0, 10000, 20260922, max(1, min(32, (os.cpu_count() or 8))))
    assert b.n_windows == 100_000
    st = _all_windows_vs_oracle(oracle, b, (3, -5, -4), "cfg3 whole, one GPU")
    assert st["n_retried"] < 100 and st["dp_cells_full"] > 1.0e12


def test_batch_larger_than_the_resident_slots(oracle):
    """9000 windows in one batch: more work items than the 2048 resident slots, so the deepest-first work queue, the
    work-item -> window indirection of the outputs and slot re-use between windows are all on the path."""
    from racon_amd.synth import simulate_windows
    b = simulate_windows(4_500_000, 500, 30.0, 10000, seed=20260927)
    assert b.n_windows == 9000
    _all_windows_vs_oracle(oracle, b, (3, -5, -4), "9000-window batch")


def test_cfg5_fragment_correction_windows_against_the_oracle(oracle):
    """BASELINE configs[4] (`-f`: the reads are the targets, dual overlaps, real backbone qualities) at 1/500 of its size:
    200 reads of 10 kbp -> ~3900 windows of ~30 layers (the full configuration is 2 M windows of the same shape)."""
    b = config_windows("cfg5", 0.002)
    assert 3000 < b.n_windows < 5000 and int(b.seq_has_qual[b.win_seq_off[:-1]].min()) == 1     # backbones carry qualities
    _all_windows_vs_oracle(oracle, b, (3, -5, -4), "cfg5 x 0.002")
    _all_windows_vs_oracle(oracle, b, (1, -1, -1), "cfg5 x 0.002, scores 1/-1/-1 (the reference's fragment tests)")


def _nccl_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from racon_amd import distributed as rd
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    b = simulate_windows(300_000, 500, 30.0, 10000, seed=20260922)
    eng = HipEngine(3, -5, -4, True, device=rank)
    out = rd.polish_sharded(b, eng.consensus, rank, world, device=torch.device("cuda", rank))
    if rank == 0:
        import hashlib
        h = hashlib.sha256()
        for c in out.consensus:
            h.update(hashlib.md5(c).digest())
        q.put((len(out.consensus), h.hexdigest()))
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_ranks_two_gpus_rccl_gather():
    """One process per GPU, shards polished independently, consensi gathered to rank 0 over RCCL (xGMI): the same bytes as
    one GPU polishing everything.  Needs two devices: skipped on a one-GPU box."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, digest = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    b = simulate_windows(300_000, 500, 30.0, 10000, seed=20260922)
    one = HipEngine(3, -5, -4, True).consensus(b)
    assert n == b.n_windows and digest == _digest(one.consensus)


def _nccl_one_rank_worker(port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import datetime
    import torch
    import torch.distributed as dist
    from racon_amd import distributed as rd
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(minutes=10))
    b = simulate_windows(100_000, 500, 30.0, 10000, seed=20260922)
    eng = HipEngine(3, -5, -4, True, device=0)
    out = rd.polish_sharded(b, eng.consensus, 0, 1, device=torch.device("cuda", 0), force_exchange=True)
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py's timing reduction
    dist.barrier()
    q.put((len(out.consensus), _digest(out.consensus), float(t.item()), dist.get_backend()))
    dist.destroy_process_group()


def test_one_rank_rccl_exchange():
    """The RCCL (`nccl`) code path on the one-GPU box: a ONE-rank process group, and through it the exchange step of
    racon_amd.distributed.polish_sharded (size all-reduce on a device tensor, slab gather to rank 0, decode) plus bench.py's
    MAX reduction and barrier -- so that the first multi-GPU run is not the first execution of that code (the two-GPU twin
    above can only skip here).  Reference: results of every device end up in one host process, src/cuda/cudapolisher.cpp:228-240."""
    import socket
    import torch.multiprocessing as mp
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(port, q))
    p.start()
    n, digest, red, backend = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0 and backend == "nccl" and red == 1.5
    b = simulate_windows(100_000, 500, 30.0, 10000, seed=20260922)
    one = HipEngine(3, -5, -4, True).consensus(b)
    assert n == b.n_windows and digest == _digest(one.consensus)


def _gloo_hip_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from racon_amd import distributed as rd
    from racon_amd.engine import HipEngine, load_library
    from racon_amd.synth import simulate_windows
    import ctypes as C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = simulate_windows(400_000, 500, 30.0, 10000, seed=20260922)
    fr, tot = C.c_uint64(), C.c_uint64()
    load_library().rcn_device_free_memory(0, C.byref(fr), C.byref(tot))
    eng = HipEngine(3, -5, -4, True, device=0, arena_bytes=int(fr.value * 0.8 / world))      # the ranks share the one device
    out = rd.polish_sharded(b, eng.consensus, rank, world)
    if rank == 0:
        q.put((len(out.consensus), _digest(out.consensus), int(out.polished.sum())))
    else:
        assert out is None
    dist.destroy_process_group()


def test_four_ranks_on_one_gpu_gather_to_rank_zero():
    """The rank-per-GPU path (racon_amd.distributed.polish_sharded: cost-balanced contiguous shards, no data-path
    collective, one gather to rank 0) with the HIP engine under every rank: four processes, four engines with a quarter of
    the arena each, on the ONE device of the box, gloo for the gather -- the bytes of one engine polishing everything."""
    import socket
    import torch.multiprocessing as mp
    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_hip_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    n, digest, npol = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    b = simulate_windows(400_000, 500, 30.0, 10000, seed=20260922)
    one = HipEngine(3, -5, -4, True).consensus(b)
    assert n == b.n_windows == 800 and digest == _digest(one.consensus) and npol == int(one.polished.sum())


def test_cfg3_one_gpu_share_through_the_product(oracle, tmp_path_factory):
    """BASELINE configs[2] at one GPU's share (50 Mbp / 8 = 6.25 Mbp, 12 500 windows) through the PRODUCT: files ->
    racon_amd/host Polisher -> initialize -> polish (three deepest-first chunks over two engines) -- the FASTA the oracle gives
    on the windows initialize() built, byte for byte; the windows themselves are the packed workload of bench.py
    (tests/test_synth_files.py).  (The whole 100 000-window job on one GPU, every window against the oracle:
    `bench.py --config cfg3 --verify`, profiles/r03/h_bench_cfg3_1gpu.json.)"""
    from racon_amd import polisher as P
    from racon_amd.synth import simulate_window_files
    P.build()
    d = str(tmp_path_factory.mktemp("cfg3share"))
    paths = simulate_window_files(d, 6_250_000, 30.0, 10000, seed=20260922, workers=16)

    def make():
        return P.Polisher(paths["reads"], paths["sam"], paths["targets"], "kC", 500, 10.0, 0.3, True, 3, -5, -4, num_threads=16)
    p = make(); p.initialize()
    b = p.windows()
    assert b.n_windows == 12500
    ref = p.assemble(oracle.consensus(b, 3, -5, -4, True, 0, simd=True), True)
    p.close()
    p = make(); p.initialize()
    got = p.polish(True)
    sec = p.polish_seconds()
    p.close()
    assert got == ref
    assert 0.0 < sec < 2.0, sec                   # (95-110 ms on an idle box; the bound only catches a product that fell off the fast path)


def test_cfg5_share_through_the_binary():
    """BASELINE configs[4] (`-f`, dual overlaps) at ONE GPU'S SHARE, 12.5 % -- 12 500 reads of 10 kbp, ~250 000 windows,
    ~600 000 overlaps to align -- through `racon_hip -f` with everything on the device: the FASTA equals the engine's
    consensus on the device-built windows for every window, a 5 % sample of those windows (12 500) equals the oracle, no
    window needed the retry pass.  (The host-aligned modes 0 / 2 against mode 3 at 1 %: tools/cfg5_at_size.py,
    profiles/r03/e_cfg5_at_size_x0.125.json; mode 0 against mode 3 at the full share, --host-at-size, 200 s of host alignment:
    profiles/r04/z_cfg5_at_size_x0.125_host_aligner.json.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "cfg5_at_size.py"), "--scale", "0.125", "--cross-scale", "0", "--threads", "16"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root).stdout
    j = json.loads(out.decode().strip().splitlines()[-1])
    assert j["device_everything"]["rc"] == 0 and 240000 < j["windows"] < 260000
    assert j["fasta_equals_engine_consensus"] is True
    assert j["oracle_sample"]["differ"] == 0 and j["oracle_sample"]["windows"] >= 12000
    assert j["consensus_kernel"]["n_retried"] == 0
