"""Multi-GPU: windows are independent (reference src/polisher.cpp:496-503; the
reference's own multi-device code hands disjoint window ranges to per-device
batch objects, src/cuda/cudapolisher.cpp:254-333), so each rank polishes a
contiguous, cost-balanced shard and there is NO collective on the data path.
The only exchange is the final variable-length gather of consensi to rank 0
(~0.6 KB/window), done with torch.distributed (RCCL over xGMI when the backend
is "nccl", gloo in the CPU tests)."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .batch import ConsensusResult, WindowBatch


def polish_sharded(batch: WindowBatch, consensus_fn: Callable[[WindowBatch], ConsensusResult], rank: int, world: int,
                   device: Optional[torch.device] = None) -> Optional[ConsensusResult]:
    """Every rank holds `batch` (or can build it); rank r polishes shard r with
    `consensus_fn` (the HIP engine in production) and rank 0 returns the
    assembled result in window order."""
    sub, idx = batch.shard(rank, world)
    res = consensus_fn(sub)
    if world == 1:
        return res
    dev = device or torch.device("cpu")
    lens = np.array([len(c) for c in res.consensus], np.int64)
    payload = np.frombuffer(b"".join(res.consensus), np.uint8)
    meta = torch.tensor([len(lens), payload.size], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    n_max = int(max(int(m[0]) for m in metas))
    p_max = int(max(int(m[1]) for m in metas))
    # equal padded slabs (one all_gather each): lengths+flags, then bytes
    head = torch.zeros(3 * n_max, dtype=torch.int64, device=dev)
    head[:len(lens)] = torch.from_numpy(lens).to(dev)
    head[n_max:n_max + len(lens)] = torch.from_numpy(res.polished.astype(np.int64)).to(dev)
    head[2 * n_max:2 * n_max + len(lens)] = torch.from_numpy(res.chimeric.astype(np.int64)).to(dev)
    body = torch.zeros(max(p_max, 1), dtype=torch.uint8, device=dev)
    if payload.size:
        body[:payload.size] = torch.from_numpy(payload.copy()).to(dev)
    heads = [torch.zeros_like(head) for _ in range(world)]
    bodies = [torch.zeros_like(body) for _ in range(world)]
    dist.all_gather(heads, head)
    dist.all_gather(bodies, body)
    if rank != 0:
        return None
    cons, pol, chi = [], [], []
    for r in range(world):
        n = int(metas[r][0])
        h = heads[r].cpu().numpy()
        bts = bodies[r].cpu().numpy().tobytes()
        off = 0
        for k in range(n):
            ln = int(h[k])
            cons.append(bts[off:off + ln]); off += ln
        pol += list(h[n_max:n_max + n]); chi += list(h[2 * n_max:2 * n_max + n])
    return ConsensusResult(cons, np.asarray(pol, np.uint8), np.asarray(chi, np.uint8))
