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
