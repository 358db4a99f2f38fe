"""N>1 path on CPU: gloo, world_size 2.  Windows shard across ranks with no
data-path collective; rank 0 gathers the variable-length consensi.  The compute
leg here is the ORACLE (tests only) — what is under test is the sharding /
gather logic of racon_amd.distributed, which is device independent."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from racon_amd.batch import WindowBatch
from racon_amd.synth import simulate_windows


def test_shard_partitions_the_window_space():
    b = simulate_windows(20000, 500, 10, 3000, seed=3)
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            sub, idx = b.shard(r, world)
            assert sub.n_windows == len(idx)
            for k, w in enumerate(idx):
                assert sub.window(k) == b.window(int(w))
            seen += list(idx)
        assert seen == list(range(b.n_windows))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from racon_amd import distributed as rd
    from oracle import oracle_lib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = simulate_windows(12000, 500, 10, 3000, seed=4)
    out = rd.polish_sharded(b, lambda sub: oracle_lib.consensus(sub, 3, -5, -4, True, 2), rank, world)
    if rank == 0:
        ref = oracle_lib.consensus(b, 3, -5, -4, True, 2)
        q.put((out.consensus == ref.consensus, bool((out.polished == ref.polished).all())))
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, flags = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same and flags


def _one_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    from racon_amd import distributed as rd
    from oracle import oracle_lib
    dist.init_process_group("gloo", rank=0, world_size=1)
    b = simulate_windows(8000, 500, 10, 3000, seed=5)
    fn = lambda sub: oracle_lib.consensus(sub, 3, -5, -4, True, 2)
    plain = rd.polish_sharded(b, fn, 0, 1)
    forced = rd.polish_sharded(b, fn, 0, 1, force_exchange=True)         # all-reduce + gather + decode on a one-rank group
    q.put((forced.consensus == plain.consensus, bool((forced.polished == plain.polished).all() and (forced.chimeric == plain.chimeric).all())))
    dist.destroy_process_group()


def test_one_rank_group_runs_the_exchange_step():
    """force_exchange: the exchange step of polish_sharded on a ONE-rank process group gives back exactly what went in (the
    RCCL twin of this test, tests/test_gpu_fullsize.py::test_one_rank_rccl_exchange, is how the nccl backend gets exercised
    on a one-GPU box)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(port, q))
    p.start()
    same, flags = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0 and same and flags


def test_cost_balanced_shards_at_cfg3_size():
    """100 000 windows (cfg3's window count), offsets only: shard bounds are monotone, cover the index space, and the cost
    proxy (layers x bases, the engine's own work-order proxy) of every shard is within 2 % of the mean although window
    depths vary 4x -- while an equal-count split of the same windows is off by much more."""
    rng = np.random.default_rng(11)
    nw = 100_000
    # coverage drifts slowly along the contig (what makes an equal-count split unbalanced)
    drift = 1.0 + 0.6 * np.sin(np.linspace(0, 3.0, nw))
    layers = np.maximum(2, rng.poisson(30 * drift)).astype(np.int64)
    win_seq_off = np.concatenate([[0], np.cumsum(layers + 1)]).astype(np.uint32)
    seq_len = rng.integers(400, 560, int(win_seq_off[-1])).astype(np.int64)
    seq_off = np.concatenate([[0], np.cumsum(seq_len)]).astype(np.uint64)
    cost = WindowBatch.window_costs(win_seq_off, seq_off)
    for world in (2, 4, 8):
        bounds = WindowBatch.shard_bounds(win_seq_off, seq_off, world)
        assert bounds[0] == 0 and bounds[-1] == nw and all(a <= b for a, b in zip(bounds, bounds[1:]))
        per = np.array([cost[bounds[r]:bounds[r + 1]].sum() for r in range(world)])
        assert per.max() / per.mean() < 1.02, (world, per / per.mean())
    equal = np.array([cost[r * nw // 8:(r + 1) * nw // 8].sum() for r in range(8)])
    assert equal.max() / equal.mean() > 1.2


def test_parallel_generator_is_the_concatenation_of_its_pieces():
    """bench.py's multi-rank default (cfg3: 50 Mbp / ranks) generates its windows as 1 Mbp stretches in worker processes:
    the result must be exactly the concatenation of simulate_windows() over the pieces' sizes and seeds, and a contig of
    one piece exactly simulate_windows() itself."""
    from racon_amd.synth import simulate_windows, simulate_windows_parallel
    b = simulate_windows_parallel(125_000, 500, 12.0, 4000, seed=77, piece=50_000, workers=3)
    parts = [simulate_windows(n, 500, 12.0, 4000, seed=77 * 1000 + k) for k, n in enumerate([50_000, 50_000, 25_000])]
    ref = parts[0].concat(parts[1]).concat(parts[2])
    assert b.n_windows == ref.n_windows == 250
    for name in ("win_seq_off", "win_type", "seq_off", "seq_has_qual", "seq_begin", "seq_end", "bases", "quals"):
        assert np.array_equal(getattr(b, name), getattr(ref, name)), name
    one = simulate_windows_parallel(40_000, 500, 12.0, 4000, seed=5, piece=50_000)
    same = simulate_windows(40_000, 500, 12.0, 4000, seed=5)
    assert np.array_equal(one.bases, same.bases) and np.array_equal(one.seq_off, same.seq_off)


def test_bench_workload_slicing():
    """What `bench.py --gpus N` runs on each rank (the driver launches it at N = 1, 2, 4, 8): cfg2 alone, else cfg3 cut into
    N equal stretches with per-rank seeds and the total fixed ("strong"); --contig gives every rank the same size ("weak")."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.pick_workload("", 0, 0, 1) == (1_000_000, 20260921, "weak", "cfg2")
    for world in (2, 4, 8):
        parts = [bench.pick_workload("", 0, r, world) for r in range(world)]
        assert sum(p[0] for p in parts) == 50_000_000 and {p[2] for p in parts} == {"strong"}
        assert [p[1] for p in parts] == [20260922 + r for r in range(world)]                 # independent stretches
        assert sum((p[0] + 499) // 500 for p in parts) == 100_000                               # cfg3's window count
    assert [bench.pick_workload("", 300_000, r, 4)[:3] for r in range(2)] == [(300_000, 20260921, "weak"), (300_000, 20260922, "weak")]
    assert bench.pick_workload("cfg3", 0, 0, 1)[:2] == (50_000_000, 20260922)
