#!/usr/bin/env python
"""bench.py — polished windows/sec of the window-consensus hot path on MI355X.

One "step" = one pass of the hot path (racon Window::generate_consensus for every
window, reference src/window.cpp:65-149 as batched at src/polisher.cpp:496-503)
over one batch of synthetic windows that is ALREADY RESIDENT IN HBM when the
timed region starts.  Workload at N=1: BASELINE.json configs[1] — synthetic
1 Mbp contig, 30x ONT-error reads, -w 500 (2000 windows).  For N>1 every rank
polishes its own equally sized shard (weak scaling; windows are independent,
there is no data-path collective: only the timing barrier/all-reduce).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contig", type=int, default=1_000_000, help="contig bp per GPU (cfg2: 1 Mbp)")
    ap.add_argument("--window", type=int, default=500)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--scores", default="3,-5,-4", help="match,mismatch,gap (racon CLI defaults, main.cpp:51-53)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=0, help="windows for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="also check every window against the oracle (untimed)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (code-path test with several ranks on ONE GPU)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.dist_backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if a.dist_backend != "nccl":
        local_rank %= torch.cuda.device_count()          # several test ranks on one GPU
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if a.dist_backend == "nccl" else "cpu"
    m, x, g = [int(v) for v in a.scores.split(",")]

    from racon_amd.engine import HipEngine
    from racon_amd.synth import simulate_windows

    batch = simulate_windows(a.contig, a.window, a.coverage, 10000, seed=20260921 + rank)
    eng = HipEngine(m, x, g, True, device=local_rank, max_slots=a.slots)
    eng.upload(batch)                                   # inputs resident in HBM from here on

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.run_only()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    for _ in range(a.steps):
        eng.run_only()                                  # kernel(s) + D2H of consensi; syncs its own stream
        st = eng.stats()
        kernel_ms += st["kernel_ms"]
        launches += st["n_launches"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nw = torch.tensor([batch.n_windows], dtype=torch.float64, device=red_dev)
        dist.all_reduce(nw, op=dist.ReduceOp.SUM)
        total_windows = int(nw.item())
    else:
        total_windows = batch.n_windows

    res = eng.result()
    st = eng.stats()
    if rank == 0:
        # --- roofline of the dominant (only) kernel: algorithmic bytes per launch / launch duration
        alg_bytes = st["dp_bytes"] + 2 * int(batch.bases.size) + 5 * sum(len(c) for c in res.consensus)
        avg_launch_s = (kernel_ms / max(1, launches)) / 1e3
        # measured HBM bytes per launch: PMC counters cannot be read from inside this process; the number comes
        # from the rocprofv3 --pmc passes of THIS command (tools/gpu_round.sh), committed as profiles/traffic.json
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and a.contig == 1_000_000 and a.window == 500 and a.coverage == 30.0:
            try:
                traffic = json.load(open(tf))["bytes_per_launch"]
            except Exception:
                traffic = None
        achieved = alg_bytes / avg_launch_s / 1e9
        out = {
            "metric": "polished windows/sec (500 bp, 30x cov)",
            "value": total_windows * a.steps / dt,
            "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": "cfg2: synthetic %d bp contig/GPU, %gx ONT-error reads (3%% sub, 3%% ins, 4%% del), -w %d, "
                                   "scores %s, %d windows/GPU" % (a.contig, a.coverage, a.window, a.scores, batch.n_windows),
                       "windows_per_gpu": batch.n_windows, "parallelism": "windows sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "poa_window_kernel2", "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "gcups": st["dp_cells"] / avg_launch_s / 1e9},
        }
        if not a.no_cpu:
            # CPU baseline on this box's host cores: the oracle's AVX2 int16 variant (the scheme of spoa's SIMD
            # engine: row vectors + log-step prefix max), all hardware threads, whole batch, best of 3.  The scalar
            # int32 oracle is timed next to it on a sample for reference.
            from oracle import oracle_lib
            cores = os.cpu_count() or 1
            oracle_lib.consensus(batch.select(range(min(64, batch.n_windows))), m, x, g, True, cores, simd=True)   # warm up
            best_dt, ref = None, None
            for _ in range(3):
                tc = time.perf_counter()
                ref = oracle_lib.consensus(batch, m, x, g, True, cores, simd=True)
                dtc = time.perf_counter() - tc
                best_dt = dtc if best_dt is None else min(best_dt, dtc)
            ok = ref.consensus == res.consensus
            n_s = a.cpu_sample or min(batch.n_windows, max(64, 4 * cores))
            sample = batch.select(range(n_s))
            tc = time.perf_counter()
            oracle_lib.consensus(sample, m, x, g, True, cores)
            dts = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": batch.n_windows / best_dt, "unit": "windows/s", "cores": cores, "kind": "port",
                                   "sample": "all %d windows of the same workload, oracle/poa_oracle.cpp AVX2 int16 variant, "
                                             "%d threads, best of 3 (%.2f s); scalar int32 oracle on the first %d windows: "
                                             "%.0f windows/s" % (batch.n_windows, cores, best_dt, n_s, n_s / dts),
                                   "matches_gpu": bool(ok)}
        if a.verify:
            from oracle import oracle_lib
            ref = oracle_lib.consensus(batch, m, x, g, True, 0)
            out["verified_windows"] = int(sum(ref.consensus[i] == res.consensus[i] for i in range(batch.n_windows)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
