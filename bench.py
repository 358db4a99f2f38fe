#!/usr/bin/env python
"""bench.py — polished windows/sec of the window-consensus hot path on MI355X.

One "step" = one pass of the hot path (racon Window::generate_consensus for every
window, reference src/window.cpp:65-149 as batched at src/polisher.cpp:496-503)
over one batch of synthetic windows that is ALREADY RESIDENT IN HBM when the
timed region starts.  Workload at N=1: BASELINE.json configs[1] (cfg2) — synthetic
1 Mbp contig, 30x ONT-error reads, -w 500 (2000 windows).  For N>1 the default is
configs[2] (cfg3): the 50 Mbp / 100 000-window job split evenly over the ranks
(rank r generates its own 50/N Mbp stretch, seed 20260922 + r; total work fixed ->
"strong"); `--contig` gives every rank that many bp instead (weak scaling).  Windows
are independent: there is no data-path collective, only the timing barrier /
all-reduce.

Next to `value` (inputs resident) the line carries `value_incl_upload`: the same
windows through pack-to-pinned + H2D + kernel + D2H, what Polisher::polish() pays per
batch (reference src/polisher.cpp:493 -> :539-543 brackets exactly that interval).
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
    ap.add_argument("--contig", type=int, default=0, help="contig bp per GPU (default: cfg2 = 1 Mbp at N=1; cfg3 = 50 Mbp / N at N>1)")
    ap.add_argument("--config", default="", help="another named workload of SURVEY 8(d) instead (cfg4, cfg5x<scale>, w1000): profiling runs, "
                                                 "not the headline metric")
    ap.add_argument("--window", type=int, default=500)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--scores", default="3,-5,-4", help="match,mismatch,gap (racon CLI defaults, main.cpp:51-53)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=0, help="windows for the CPU baseline (0 = auto)")
    ap.add_argument("--cpu-threads", default="", help="comma list of thread counts for the CPU baseline sweep (default: 32,64,128,all)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the untimed pack + upload + run leg (value_incl_upload): the rocprofv3 passes "
                                                                 "use it so that every traced launch of the kernel is one of the timed whole-batch launches")
    ap.add_argument("--verify", action="store_true", help="also check every window against the oracle (untimed)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (code-path test with several ranks on ONE GPU)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    m, x, g = [int(v) for v in a.scores.split(",")]

    from racon_amd.engine import HipEngine
    from racon_amd.synth import config_windows, simulate_windows, simulate_windows_parallel

    # The synthetic input first: long contigs are generated in forked worker processes, and nothing of HIP / RCCL (contexts,
    # helper threads) may exist in the parent when it forks.  Rank and world size come from the launcher's environment.
    if a.config:
        scaling, cfg_name = "weak", a.config
        if a.config.startswith("cfg5x"):
            batch = config_windows("cfg5", float(a.config[5:]))
        else:
            batch = config_windows(a.config)
        contig = 0
    elif a.contig:
        contig, seed, scaling, cfg_name = a.contig, 20260921 + rank, "weak", "cfg2-shaped"
    elif world == 1:
        contig, seed, scaling, cfg_name = 1_000_000, 20260921, "weak", "cfg2"
    else:
        contig, seed, scaling, cfg_name = 50_000_000 // world, 20260922 + rank, "strong", "cfg3 (50 Mbp / %d ranks)" % world
    a.contig = contig
    if not a.config:
        # (long contigs are generated as 1 Mbp stretches in worker processes before any GPU work starts: the generator
        #  is a Python loop per read, 8 s per Mbp; a 1 Mbp contig is exactly simulate_windows(...))
        workers = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
        batch = simulate_windows_parallel(contig, a.window, a.coverage, 10000, seed=seed, workers=workers) if contig > 1_000_000 \
            else simulate_windows(contig, a.window, a.coverage, 10000, seed=seed)
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
    # the same windows including pack + upload (what the product's polish() pays per batch); outside the timed region above
    dt_up = float("nan")
    if not a.no_upload_leg:
        eng.consensus(batch)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.consensus(batch)
        barrier()
        dt_up = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_up], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_up = float(t.item())
    st_up = eng.stats()
    if rank == 0:
        # --- roofline of the dominant (only) kernel: algorithmic bytes per launch / launch duration
        alg_bytes = st["dp_bytes"] + 2 * int(batch.bases.size) + 5 * sum(len(c) for c in res.consensus)
        avg_launch_s = (kernel_ms / max(1, launches)) / 1e3
        # measured HBM bytes per launch: PMC counters cannot be read from inside this process; the number comes
        # from the rocprofv3 --pmc passes of THIS command (tools/gpu_round.sh -> tools/pmc_summary.py), committed as
        # profiles/traffic.json and stamped with the hash of the kernel sources it was measured on: a stale file
        # (sources changed since) is reported as null, not quoted
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and world == 1 and a.contig == 1_000_000 and a.window == 500 and a.coverage == 30.0:
            try:
                tj = json.load(open(tf))
                from tools.srchash import kernel_source_hash
                if tj.get("kernel_source_hash") == kernel_source_hash():
                    traffic, traffic_src = tj["bytes_per_launch"], "profiles/traffic.json (%s)" % tj.get("measured", "?")
                else:
                    traffic_src = "profiles/traffic.json is stale (kernel sources changed since it was measured)"
            except Exception:
                traffic = None
        achieved = alg_bytes / avg_launch_s / 1e9
        out = {
            "metric": "polished windows/sec (500 bp, 30x cov)",
            "value": total_windows * a.steps / dt,
            "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "value_incl_upload": None if a.no_upload_leg else total_windows * a.steps / dt_up,
            "ms_per_step_incl_upload": None if a.no_upload_leg else dt_up / a.steps * 1e3,
            "upload": {"h2d_ms": st_up["h2d_ms"], "d2h_ms": st_up["d2h_ms"], "bytes_in": st_up["bytes_in"]},
            "config": {"workload": "%s: synthetic %d bp contig/GPU, %gx ONT-error reads (3%% sub, 3%% ins, 4%% del), -w %d, "
                                   "scores %s, %d windows/GPU" % (cfg_name, a.contig, a.coverage, a.window, a.scores, batch.n_windows),
                       "windows_per_gpu": batch.n_windows, "parallelism": "windows sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "poa_window_kernel2", "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "gcups": st["dp_cells"] / avg_launch_s / 1e9,
                         # exact banded DP ("cells counts the cells actually evaluated and the full-matrix figure
                         # is reported alongside"): the figures an unbanded pass over the same alignments has
                         "full_matrix": {"algorithmic_bytes_per_launch": alg_bytes - st["dp_bytes"] + st["dp_bytes_full"],
                                         "achieved": (alg_bytes - st["dp_bytes"] + st["dp_bytes_full"]) / avg_launch_s / 1e9,
                                         "gcups": st["dp_cells_full"] / avg_launch_s / 1e9},
                         "banded_alignments": st["n_banded"], "band_redone": st["n_band_redone"], "band_redo_why": st["band_redo_why"],
                         "phase_clocks": st["phase_clocks"], "work_groups_per_cu": st["wg_per_cu"]},
        }
        if not a.no_cpu and world == 1:
            # CPU baseline on this box's host cores (rank 0 at N = 1 only): the oracle's AVX2 int16 variant (the scheme of spoa's SIMD
            # engine: row vectors + log-step prefix max), all hardware threads, whole batch, best of 3.  The scalar
            # int32 oracle is timed next to it on a sample for reference.
            from oracle import oracle_lib
            ncpu = os.cpu_count() or 1
            # bounded sample of the same workload: at most 2000 windows (all of cfg2)
            cb = batch if batch.n_windows <= 2000 else batch.select(range(2000))
            want = [int(v) for v in a.cpu_threads.split(",") if v] or [32, 64, 128, ncpu]
            sweep_threads = sorted({min(max(1, v), ncpu) for v in want})
            oracle_lib.consensus(cb.select(range(min(64, cb.n_windows))), m, x, g, True, ncpu, simd=True)   # warm up
            sweep, best_dt, ref, cores = {}, None, None, ncpu
            for th in sweep_threads:                       # the box's best thread count is what the GPU is compared with
                bt = None
                for _ in range(3):
                    tc = time.perf_counter()
                    ref = oracle_lib.consensus(cb, m, x, g, True, th, simd=True)
                    dtc = time.perf_counter() - tc
                    bt = dtc if bt is None else min(bt, dtc)
                sweep[str(th)] = cb.n_windows / bt
                if best_dt is None or bt < best_dt:
                    best_dt, cores = bt, th
            ok = ref.consensus == res.consensus[:cb.n_windows]
            n_s = a.cpu_sample or min(cb.n_windows, max(64, 4 * ncpu))
            sample = cb.select(range(n_s))
            tc = time.perf_counter()
            oracle_lib.consensus(sample, m, x, g, True, ncpu)
            dts = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": cb.n_windows / best_dt, "unit": "windows/s", "cores": cores, "kind": "port",
                                   "sample": "%d windows of the same workload, oracle/poa_oracle.cpp AVX2 int16 variant, best thread "
                                             "count of the sweep (%d of %d hardware threads), best of 3 (%.2f s); scalar int32 oracle "
                                             "on the first %d windows, all threads: %.0f windows/s"
                                             % (cb.n_windows, cores, ncpu, best_dt, n_s, n_s / dts),
                                   "thread_sweep_windows_per_s": sweep,
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
