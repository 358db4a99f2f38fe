"""Multi-GPU: windows are independent (reference src/polisher.cpp:496-503; the
reference's own multi-device code hands disjoint window ranges to per-device
batch objects, src/cuda/cudapolisher.cpp:254-333), so each rank polishes a
contiguous, cost-balanced shard and there is NO collective on the data path.
The only exchange is the final variable-length gather of consensi to rank 0
(~0.6 KB/window; 60 MB for the 100 000 windows of cfg3), done with
torch.distributed (RCCL over xGMI when the backend is "nccl", gloo in the CPU
tests): a gather TO RANK 0, not an all-gather -- nobody else needs the bytes."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .batch import ConsensusResult, WindowBatch


def polish_sharded(batch: WindowBatch, consensus_fn: Callable[[WindowBatch], ConsensusResult], rank: int, world: int,
                   device: Optional[torch.device] = None, force_exchange: bool = False) -> Optional[ConsensusResult]:
    """Every rank holds `batch` (or can build it); rank r polishes shard r with
    `consensus_fn` (the HIP engine in production) and rank 0 returns the
    assembled result in window order (the other ranks return None).

    The one exchange step: a 16-byte all-reduce (MAX) so that all ranks agree on the slab size, then ONE gather of
    equal padded slabs to rank 0 ({count, bytes, lengths, flags} as int64 + the consensus bytes) -- rank 0's links
    carry (world - 1) slabs, nobody else receives anything (reference analogue: results of every device's batches end
    up in the one host process, src/cuda/cudapolisher.cpp:305-308).

    `force_exchange`: run the all-reduce and the gather even when world == 1 (a one-rank process group) -- the RCCL code
    path on a one-GPU box: tests/test_gpu_fullsize.py::test_one_rank_rccl_exchange, bench.py's `rccl_selftest`."""
    sub, idx = batch.shard(rank, world)
    res = consensus_fn(sub)
    if world == 1 and not force_exchange:
        return res
    dev = device or torch.device("cpu")
    n = len(res.consensus)
    payload = np.frombuffer(b"".join(res.consensus), np.uint8)
    size = torch.tensor([n, payload.size], dtype=torch.int64, device=dev)
    dist.all_reduce(size, op=dist.ReduceOp.MAX)
    n_max, p_max = int(size[0]), int(size[1])
    head_words = 2 + 3 * n_max
    slab = torch.zeros(8 * head_words + max(p_max, 1), dtype=torch.uint8, device=dev)
    head = np.zeros(head_words, np.int64)
    head[0], head[1] = n, payload.size
    head[2:2 + n] = [len(c) for c in res.consensus]
    head[2 + n_max:2 + n_max + n] = res.polished
    head[2 + 2 * n_max:2 + 2 * n_max + n] = res.chimeric
    slab[:8 * head_words] = torch.from_numpy(head.view(np.uint8).copy()).to(dev)
    if payload.size:
        slab[8 * head_words:8 * head_words + payload.size] = torch.from_numpy(payload.copy()).to(dev)
    slabs = [torch.zeros_like(slab) for _ in range(world)] if rank == 0 else None
    dist.gather(slab, gather_list=slabs, dst=0)
    if rank != 0:
        return None
    cons, pol, chi = [], [], []
    for r in range(world):
        raw = slabs[r].cpu().numpy()
        h = raw[:8 * head_words].view(np.int64)
        nr = int(h[0])
        bts = raw[8 * head_words:8 * head_words + int(h[1])].tobytes()
        off = 0
        for k in range(nr):
            ln = int(h[2 + k])
            cons.append(bts[off:off + ln]); off += ln
        pol += list(h[2 + n_max:2 + n_max + nr]); chi += list(h[2 + 2 * n_max:2 + 2 * n_max + nr])
    return ConsensusResult(cons, np.asarray(pol, np.uint8), np.asarray(chi, np.uint8))
