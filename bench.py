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

WHICH NUMBER IS WHICH (round 6).  On the default workload (cfg2 at N = 1) `value` is THE CONTRACT METRIC of SURVEY.md 8(d): polished
windows/s on the product's Polisher::polish() interval (value_product_polish, in-process).  `value_kernel_leg` is the timed loop of
this file -- K steps over inputs resident in HBM, windows x K / (K x ms_per_step) -- and it is what `ms_per_step`, `steps` and
`roofline` describe.  Wherever the product leg does not run (N > 1, --no-product, other workloads) `value` is the kernel leg;
`value_is` says which.

Next to them the line carries
  value_incl_upload     the same windows through pack-to-pinned + H2D + kernel + D2H on a warm engine;
  value_product_polish  THE PRODUCT: the same workload written as racon input files (FASTQ + SAM + FASTA), read by
                        racon_amd/host's Polisher (include/racon_host.h), and the interval the reference's Logger
                        brackets around Polisher::polish (reference src/polisher.cpp:493 -> :539-543) -- the interval
                        SURVEY.md 8(d) defines "polished windows / second" on -- for cfg2's 2000 windows and for one
                        GPU's share of cfg3 (6.25 Mbp, 12 500 windows); the FASTA is checked against the kernel leg.
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
CACHE = os.environ.get("RACON_AMD_CACHE", "/tmp/racon_amd_cache")


def cached_windows(contig, window, coverage, read_len, seed, workers):
    """simulate_windows_parallel, kept as .npz in a scratch directory keyed by its arguments: the generator is a Python
    loop per read (8 s per Mbp), and the driver runs this file at N = 1, 2, 4, 8 back to back on one box."""
    from racon_amd.batch import WindowBatch
    from racon_amd.synth import simulate_windows, simulate_windows_parallel
    key = os.path.join(CACHE, "win_%d_%d_%g_%d_%d.npz" % (contig, window, coverage, read_len, seed))
    if os.path.exists(key):
        try:
            return WindowBatch.load(key)
        except Exception:
            pass
    b = simulate_windows_parallel(contig, window, coverage, read_len, seed=seed, workers=workers) if contig > 1_000_000 \
        else simulate_windows(contig, window, coverage, read_len, seed=seed)
    if contig >= 1_000_000:
        try:
            os.makedirs(CACHE, exist_ok=True)
            np.savez(key + ".tmp.npz", win_seq_off=b.win_seq_off, win_type=b.win_type, seq_off=b.seq_off, seq_has_qual=b.seq_has_qual,
                     seq_begin=b.seq_begin, seq_end=b.seq_end, bases=b.bases, quals=b.quals)
            os.replace(key + ".tmp.npz", key)
        except Exception:
            pass
    return b


def cpu_limits():
    """What this process may use of the host: the numbers the CPU baseline's "all host cores" has to be read against."""
    out = {"os_cpu_count": os.cpu_count(), "sched_affinity": len(os.sched_getaffinity(0))}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out[f.rsplit("/", 1)[1]] = open(f).read().strip()
        except Exception:
            pass
    try:
        out["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except Exception:
        pass
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        out["cpu_model"] = model[0] if model else None
    except Exception:
        pass
    q = out.get("cpu.max", "max").split()
    out["cgroup_cpus"] = (float(q[0]) / float(q[1])) if len(q) == 2 and q[0] != "max" else None
    return out


def physical_cores():
    """Physical cores of the box (unique (package, core) pairs of /proc/cpuinfo), whatever the lease lets this process use."""
    try:
        cores, pkg = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cores.add((pkg, line.split(":", 1)[1].strip()))
        return len(cores) or None
    except Exception:
        return None


def rccl_selftest(eng, batch, local_rank):
    """N = 1: the RCCL (`nccl`) code path of the multi-GPU line on a ONE-rank process group -- init, the size all-reduce and slab
    gather of racon_amd.distributed.polish_sharded (device tensors), the timing MAX reduction and the barrier -- so that the first
    run on several GPUs is not the first execution of that code.  Untimed; the result goes into the line."""
    import datetime
    import socket
    from racon_amd import distributed as rd
    t0 = time.perf_counter()
    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=5))
        sub = batch.select(range(min(256, batch.n_windows)))
        plain = eng.consensus(sub)
        forced = rd.polish_sharded(sub, eng.consensus, 0, 1, device=torch.device("cuda", local_rank), force_exchange=True)
        t = torch.tensor([2.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        ok = forced.consensus == plain.consensus and bool((forced.polished == plain.polished).all()) and float(t.item()) == 2.5
        backend = dist.get_backend()
        dist.destroy_process_group()
        return {"ok": bool(ok), "backend": backend, "world": 1, "windows": sub.n_windows, "seconds": round(time.perf_counter() - t0, 2),
                "what": "one-rank RCCL group: polish_sharded(force_exchange) all-reduce + gather on device tensors, MAX reduction, barrier"}
    except Exception as e:
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        return {"ok": False, "error": repr(e)[:300]}


def product_files(contig, coverage, seed, workers, short_reads=False):
    """The workload as racon input files in the scratch cache (generated in forked workers: call before HIP exists).
    `short_reads`: the cfg4 shape (150-base reads, 0.3 % substitutions, 0.05 % insertions / deletions, phred 30)."""
    from racon_amd.synth import simulate_window_files
    d = os.path.join(CACHE, "files_%d_%g_%d%s" % (contig, coverage, seed, "_short" if short_reads else ""))
    t0 = time.perf_counter()
    if not all(os.path.exists(os.path.join(d, n)) for n in ("targets.fasta", "reads.fastq", "overlaps.sam", ".done")):
        if short_reads:
            simulate_window_files(d, contig, coverage, 150, seed=seed, piece=125_000, workers=workers, sub=0.003, ins=0.0005, dele=0.0005, phred=(30.0, 0.0, 30, 30))
        else:
            simulate_window_files(d, contig, coverage, 10000, seed=seed, workers=workers)
        open(os.path.join(d, ".done"), "w").close()
    return {"targets": os.path.join(d, "targets.fasta"), "reads": os.path.join(d, "reads.fastq"), "sam": os.path.join(d, "overlaps.sam"),
            "paf": os.path.join(d, "overlaps.paf"), "contig": contig, "files_s": round(time.perf_counter() - t0, 1)}


def product_polish(paths, window, scores, threads, expect=None, reps=2, batches=1, mode="auto"):
    """files -> racon_amd.host Polisher -> initialize() -> polish(); returns the Logger-bracketed polish() interval.
    `mode`: RACON_HIP_DEVICE_WINDOWS for this leg -- "auto" is what the binary does (windows built in HBM at the end of
    initialize() when the job fits: polish() is the consensus alone), "0" the host-built path (windows packed and uploaded
    chunk by chunk INSIDE polish(), as the reference's GPU path does it, src/cuda/cudapolisher.cpp:254-276)."""
    from racon_amd.polisher import Polisher
    m, x, g = scores
    best, runs, nw, same = None, [], 0, None
    # (the library keeps host-built windows by default -- its callers may ask for windows() --; the product, `racon_hip`, builds them in
    #  HBM at the end of initialize() whenever they fit: the same here)
    # (set for this leg only and restored: later legs and the subprocesses they start must not inherit it)
    saved = os.environ.get("RACON_HIP_DEVICE_WINDOWS")
    os.environ["RACON_HIP_DEVICE_WINDOWS"] = mode
    try:
        for _ in range(reps):
            p = Polisher(paths["reads"], paths["sam"], paths["targets"], "kC", window, 10.0, 0.3, True, m, x, g, threads, batches)
            t1 = time.perf_counter()
            p.initialize()
            t_init = time.perf_counter() - t1
            nw = p.num_windows()
            fasta = p.polish(True)
            sec = p.polish_seconds()
            p.close()
            runs.append({"initialize_s": round(t_init, 3), "polish_s": round(sec, 5)})
            best = sec if best is None else min(best, sec)
            if expect is not None:
                same = b"".join(fasta.split(b"\n")[1::2]) == expect
    finally:
        if saved is None:
            os.environ.pop("RACON_HIP_DEVICE_WINDOWS", None)
        else:
            os.environ["RACON_HIP_DEVICE_WINDOWS"] = saved
    # (next to the polish() interval: the same rate with initialize() in it -- the reference's GPU path allocates its batches inside
    #  polish(), src/cuda/cudapolisher.cpp:212-242; here engines, arenas and pinned staging are set up behind the file parsing in
    #  initialize(), like the CPU path's Prealloc in the constructor, src/polisher.cpp:176-183)
    incl_init = min(r["initialize_s"] + r["polish_s"] for r in runs)
    return {"windows": nw, "polish_s": best, "windows_per_s": nw / best, "windows_per_s_incl_initialize": nw / incl_init, "runs": runs, "files_s": paths["files_s"],
            "fasta_matches_kernel_leg": same, "device_windows_mode": mode,
            "what": "racon_amd/host Polisher on %d bp of cfg-shaped files (FASTQ + SAM + FASTA), -t %d: the interval of reference "
                    "src/polisher.cpp:493 -> :539-543 (rcnh_polisher_polish_seconds); best of %d createPolisher + initialize + polish rounds"
                    % (paths["contig"], threads, reps)}


def product_cli(paths, window, scores, threads, expect=None, reps=2, batches=1, overlaps="sam", flags=(), env_add=None):
    """The same through the drop-in BINARY (racon_amd/host/racon_hip, the reference's command line): the interval is the
    Logger's own line, "[racon::Polisher::polish] generated consensus <s> s" (reference src/polisher.cpp:539-543).
    `overlaps`: which overlap file ("sam": CIGARs in the file; "paf": none, the overlaps are aligned first -- reference
    src/overlap.cpp:205-224); `flags` / `env_add`: the device-side modes (--cudaaligner-batches 1 = alignment, CIGAR
    walk and window construction in HBM; RACON_HIP_DEVICE_WINDOWS=2 = CIGAR walk and window construction in HBM)."""
    import hashlib
    import re
    import subprocess
    m, x, g = scores
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    best, runs, same, md5, best_wall = None, [], None, None, None
    env = dict(os.environ)
    env.pop("RACON_HIP_DEVICE_WINDOWS", None)
    env.update(env_add or {})
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-t", str(threads), "-w", str(window), "-m", str(m), "-x", str(x), "-g", str(g), "-c", str(batches)] + list(flags) +
                           [paths["reads"], paths[overlaps], paths["targets"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        wall = time.perf_counter() - t0
        best_wall = wall if best_wall is None else min(best_wall, wall)
        md5 = hashlib.md5(r.stdout).hexdigest()
        mt = re.search(r"\[racon::Polisher::polish\] generated consensus (\d+\.\d+) s", r.stderr.decode(errors="replace"))
        if r.returncode != 0 or not mt:
            return {"error": "racon_hip exit %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-200:])}
        sec = float(mt.group(1))
        runs.append({"wall_s": round(wall, 3), "polish_s": sec})
        best = sec if best is None else min(best, sec)
        if expect is not None:
            same = b"".join(r.stdout.split(b"\n")[1::2]) == expect
    nw = sum((int(l) + window - 1) // window for l in [len(x) for x in open(paths["targets"], "rb").read().split(b"\n")[1::2]])
    return {"windows": nw, "polish_s": best, "windows_per_s": nw / best, "wall_s": round(best_wall, 3), "windows_per_s_whole_binary": nw / best_wall,
            "runs": runs, "fasta_matches_kernel_leg": same, "fasta_md5": md5}


def product_multi_device(paths, window, scores, threads, n_devices, fake=False):
    """THE PRODUCT ON SEVERAL DEVICES: one racon_hip process drives every visible device (racon_amd/host/polisher.cpp: two
    engines per device pulling deepest-first chunks off one cursor -- the reference's organisation, src/cuda/cudapolisher.cpp:
    228-240, 254-276), against the same files on ONE device: polish() windows/s of both and whether the FASTA is the same.
    `fake`: RACON_HIP_FAKE_DEVICES (several logical devices on one GPU: the code path, not a speed-up)."""
    import hashlib
    import re
    import subprocess
    m, x, g = scores
    exe = os.path.join(ROOT, "racon_amd", "host", "racon_hip")
    out = {"devices": n_devices, "fake_devices": bool(fake)}
    md5 = {}
    for label, env_add in (("one_device", {"RACON_HIP_FAKE_DEVICES": "1"} if fake else {"HIP_VISIBLE_DEVICES": "0"}),
                           ("all_devices", {"RACON_HIP_FAKE_DEVICES": str(n_devices)} if fake else {})):
        env = dict(os.environ)
        for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "RACON_HIP_FAKE_DEVICES"):
            env.pop(k, None)
        env.update(env_add)
        best = None
        for _ in range(2):
            r = subprocess.run([exe, "-t", str(threads), "-w", str(window), "-m", str(m), "-x", str(x), "-g", str(g),
                                paths["reads"], paths["sam"], paths["targets"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            mt = re.search(r"\[racon::Polisher::polish\] generated consensus (\d+\.\d+) s", r.stderr.decode(errors="replace"))
            if r.returncode != 0 or not mt:
                return {"error": "racon_hip (%s) exit %d: %s" % (label, r.returncode, r.stderr.decode(errors="replace")[-300:])}
            sec = float(mt.group(1))
            best = sec if best is None else min(best, sec)
            md5[label] = hashlib.md5(r.stdout).hexdigest()
        nw = sum((int(l) + window - 1) // window for l in [len(t) for t in open(paths["targets"], "rb").read().split(b"\n")[1::2]])
        out[label] = {"polish_s": best, "windows_per_s": nw / best, "windows": nw, "fasta_md5": md5[label]}
    out["fasta_identical"] = md5["one_device"] == md5["all_devices"]
    out["speedup"] = out["all_devices"]["windows_per_s"] / out["one_device"]["windows_per_s"]
    return out


def pick_workload(config: str, contig: int, rank: int, world: int):
    """(contig bp on this rank, seed, scaling, name) of the seeded ONT-like workloads:
    N = 1: cfg2 (1 Mbp, seed 20260921).  N > 1: cfg3, the 50 Mbp / 100 000-window job cut into N equal stretches, rank r
    generating its own (seed 20260922 + r): total work is fixed -> "strong".  --contig: that many bp on EVERY rank -> "weak".
    --config cfg3: the whole 50 Mbp job on this rank's one GPU."""
    if config == "cfg3":
        return 50_000_000, 20260922, "weak", "cfg3 (whole job on one GPU)"
    if contig:
        return contig, 20260921 + rank, "weak", "cfg2-shaped"
    if world == 1:
        return 1_000_000, 20260921, "weak", "cfg2"
    return 50_000_000 // world, 20260922 + rank, "strong", "cfg3 (50 Mbp / %d ranks)" % world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contig", type=int, default=0, help="contig bp per GPU (default: cfg2 = 1 Mbp at N=1; cfg3 = 50 Mbp / N at N>1)")
    ap.add_argument("--config", default="", help="another named workload of SURVEY 8(d) instead (cfg3, cfg4, cfg5x<scale>, w1000): profiling runs, "
                                                 "not the headline metric")
    ap.add_argument("--window", type=int, default=500)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--scores", default="3,-5,-4", help="match,mismatch,gap (racon CLI defaults, main.cpp:51-53)")
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=0, help="windows for the CPU baseline (0 = auto)")
    ap.add_argument("--cpu-threads", default="", help="comma list of thread counts for the CPU baseline sweep (default: 8,16,24,32,48,64,96,128,all)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the untimed pack + upload + run leg (value_incl_upload): the rocprofv3 passes "
                                                                 "use it so that every traced launch of the kernel is one of the timed whole-batch launches")
    ap.add_argument("--no-product", action="store_true", help="skip the product leg (files -> Polisher::polish)")
    ap.add_argument("--no-device-modes", action="store_true", help="product leg: skip the device-side construction / alignment modes (SURVEY 8(f))")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="N = 1: skip the one-rank RCCL exchange self-test (untimed)")
    ap.add_argument("--product-contig", type=int, default=6_250_000, help="second product job: contig bp (one GPU's share of cfg3)")
    ap.add_argument("--product-batches", type=int, default=1, help="-c of the product legs: batch objects (pairs of engines) per device")
    ap.add_argument("--verify", action="store_true", help="also check every window against the oracle (untimed)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (code-path test with several ranks on ONE GPU)")
    ap.add_argument("--product-multi-contig", type=int, default=0, help="N > 1: contig bp of the product leg that ONE racon_hip process polishes on all N devices "
                                                                         "(default: 2 Mbp per device; 50000000 = cfg3 whole)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` without a launcher: one rank per GPU through torch.distributed.run, as the driver does it
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a line for the wrong N" % (a.gpus, world))
    m, x, g = [int(v) for v in a.scores.split(",")]

    from racon_amd.engine import HipEngine
    from racon_amd.synth import config_windows

    # The synthetic input first: long contigs are generated in forked worker processes, and nothing of HIP / RCCL (contexts,
    # helper threads) may exist in the parent when it forks.  Rank and world size come from the launcher's environment.
    workers = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    seed = None
    if a.config and a.config != "cfg3":
        scaling, cfg_name = "weak", a.config
        if a.config.startswith("cfg5x"):
            batch = config_windows("cfg5", float(a.config[5:]))
            what = "%s: fragment correction (-f), %g of 100 000 x 10 kbp ONT-error reads with dual overlaps, -w %d" % (a.config, float(a.config[5:]), 500)
        else:
            batch = config_windows(a.config)
            what = {"cfg4": "cfg4: synthetic 1 Mbp contig, 60x of 150 bp short reads (0.3% sub, 0.05% ins, 0.05% del, phred 30), -w 200, kNGS",
                    "w1000": "w1000: synthetic 1 Mbp contig, 30x ONT-error reads, -w 1000",
                    "ngs_w500": "ngs_w500: synthetic 1 Mbp contig, 60x of 150 bp short reads, -w 500 (racon's default window), kNGS"}.get(a.config, a.config)
        contig = 0
    else:
        contig, seed, scaling, cfg_name = pick_workload(a.config, a.contig, rank, world)
        batch = cached_windows(contig, a.window, a.coverage, 10000, seed, workers)
        what = "%s: synthetic %d bp contig/GPU, %gx ONT-error reads (3%% sub, 3%% ins, 4%% del), -w %d" % (cfg_name, contig, a.coverage, a.window)
    a.contig = contig
    # the product leg's input files (same seeds -> the same windows as the packed batches; generated in forked workers too)
    do_product = not a.no_product and world == 1 and ((a.window == 500 and ((not a.config and contig == 1_000_000) or a.config == "cfg3")) or a.config == "cfg4")
    pfiles = []
    pwindow = 200 if a.config == "cfg4" else a.window
    if do_product and a.config == "cfg4":
        # the same SHAPE as files (1 Mbp, 150-base reads at 60x, -w 200): the product's polish() interval on short reads
        pfiles.append(("cfg4_files", product_files(1_000_000, 60.0, 20260923, workers, short_reads=True)))
    elif do_product and a.config == "cfg3":
        pfiles.append(("cfg3", product_files(50_000_000, a.coverage, 20260922, workers)))       # the whole 100 000-window job as files
    elif do_product:
        pfiles.append(("cfg2", product_files(1_000_000, a.coverage, 20260921, workers)))
        if a.product_contig > 1_000_000:
            pfiles.append(("cfg3_share", product_files(a.product_contig, a.coverage, 20260922, workers)))
    pmulti = None
    if world > 1 and not a.no_product and a.window == 500 and not a.config and rank == 0:
        # 2 Mbp (4000 windows) per device by default: 16 Mbp of files at N = 8 are written and polished twice within the
        # minutes the default line may take (cfg3 whole -- 50 Mbp, 7 GB of text -- with --product-multi-contig 50000000)
        pmulti = product_files(a.product_multi_contig or min(50_000_000, 2_000_000 * world), a.coverage, 20260922, workers)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (an explicit, long timeout: rank 0 runs two racon_hip jobs of the product leg while the other ranks sit at the last
        #  barrier -- the default watchdog of the nccl backend, 10 min, must not be what ends a cfg3-sized product leg)
        import datetime
        tmo = datetime.timedelta(minutes=90)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(a.dist_backend, timeout=tmo)
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
    kernel_ms, launches, launch_ms = 0.0, 0, [0.0, 0.0]
    for _ in range(a.steps):
        eng.run_only()                                  # kernel(s) + D2H of consensi; syncs its own stream
        st = eng.stats()
        kernel_ms += st["kernel_ms"]                    # the interval the (one, or two concurrent) launches of a step cover
        launches += st["n_launches"]
        launch_ms = [launch_ms[k] + st["launch_ms"][k] for k in range(2)]
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
    # the same windows including pack + upload (a warm engine's polish call); outside the timed region above
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
        # --- roofline of the dominant (only) kernel: algorithmic bytes per step / the interval its launches cover.
        # A step is ONE launch of poa_window_kernel2, or -- split launch, rcn_run_stats.split_deep > 0 -- TWO CONCURRENT
        # launches of it on disjoint CU sets (deepest windows / all others): `step_kernel_ms` is the interval they
        # cover together (HIP events, first begin to last end), `launch_ms` their individual durations (what rocprofv3's
        # per-dispatch statistics average over); the bytes are the step's, so achieved = bytes per step / step interval.
        alg_bytes = st["dp_bytes"] + 2 * int(batch.bases.size) + 5 * sum(len(c) for c in res.consensus)
        step_s = (kernel_ms / a.steps) / 1e3
        per_launch_ms = [v / a.steps for v in launch_ms] if st["split_deep"] else [kernel_ms / max(1, launches)]
        # measured HBM bytes per step: PMC counters cannot be read from inside this process; the number comes
        # from the rocprofv3 --pmc passes of THIS command (tools/gpu_round.sh -> tools/pmc_summary.py), committed as
        # profiles/traffic.json and stamped with the hash of the kernel sources it was measured on: a stale file
        # (sources changed since) is reported as null, not quoted
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.config if a.config else "traffic.json")
        if os.path.exists(tf) and world == 1 and (a.config or (a.contig == 1_000_000 and a.window == 500 and a.coverage == 30.0)):
            try:
                tj = json.load(open(tf))
                from tools.srchash import kernel_source_hash
                if tj.get("kernel_source_hash") == kernel_source_hash():
                    traffic, traffic_src = tj["bytes_per_launch"], "profiles/%s (%s)" % (os.path.basename(tf), tj.get("measured", "?"))
                else:
                    traffic_src = "profiles/%s is stale (kernel sources changed since it was measured)" % os.path.basename(tf)
            except Exception:
                traffic = None
        # secondary ceilings ("the kernel may be issue-bound before it is HBM-bound"): vector / scalar issue and
        # LDS-array occupancy of the dominant kernel from the rocprofv3 SQ-counter passes of THIS command (tools/gpu_round4.sh sq
        # -> tools/sq_summary.py), committed as profiles/issue_<workload>.json and stamped like traffic.json
        secondary = None
        wl = a.config if a.config else ("cfg2" if (world == 1 and a.contig == 1_000_000 and a.window == 500 and a.coverage == 30.0) else None)
        jf = os.path.join(ROOT, "profiles", "issue_%s.json" % wl) if wl else None
        if jf and os.path.exists(jf):
            try:
                ij = json.load(open(jf))
                from tools.srchash import kernel_source_hash
                if ij.get("kernel_source_hash") == kernel_source_hash():
                    ks = ij["kernels"]
                    dom = max(ks, key=lambda k: ks[k].get("wave_instructions", 0))
                    secondary = {k: ks[dom].get(k) for k in ("valu_issue_frac", "scalar_issue_frac", "lds_frac", "lds_conflict_share", "wave_time_issuing",
                                                            "wave_time_parked_waitcnt", "wave_time_issue_stalled", "cycles_per_instruction_per_simd")}
                    secondary.update({"kernel": dom, "source": "profiles/issue_%s.json (%s)" % (wl, ij.get("measured", "?")),
                                      "note": "valu/scalar_issue_frac = SQ_ACTIVE_INST_* x 4 / (dispatch cycles x 1024 SIMDs); lds_frac = SQ_LDS_IDX_ACTIVE / (dispatch cycles x 256 CUs)"})
                else:
                    secondary = {"source": "profiles/issue_%s.json is stale (kernel sources changed since it was measured)" % wl}
            except Exception:
                secondary = None
        achieved = alg_bytes / step_s / 1e9
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
            "config": {"workload": "%s, scores %s, %d windows/GPU" % (what, a.scores, batch.n_windows),
                       "windows_per_gpu": batch.n_windows, "parallelism": "windows sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "secondary": secondary,
                         "kernel": ("poa_window_kernel_small (+ poa_window_kernel2 for the windows it sent back)" if st["n_small"] else "poa_window_kernel2") + (" + poa_window_kernel2_deep (the instance for the deep launch)" if st["split_deep"] else ""), "step_kernel_ms": step_s * 1e3, "avg_launch_ms": sum(per_launch_ms) / len(per_launch_ms),
                         "launches_per_step": len(per_launch_ms), "launch_ms": per_launch_ms,
                         "split_launch": None if not st["split_deep"] else
                             {"deep_windows": st["split_deep"], "deep_cus": st["split_cus"], "deep_work_groups_per_cu": st["split_deep_per_cu"],
                              "mid_windows": st["split_mid"], "mid_cus": st["split_mid_cus"], "mid_work_groups_per_cu": st["split_mid_per_cu"],
                              "mid_launch_ms": st["launch_ms_mid"] if st["split_mid"] else None,
                              "note": "two (three with a middle tier) concurrent launches on disjoint CU sets; bytes and interval are the step's"},
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "gcups": st["dp_cells"] / step_s / 1e9,
                         # exact banded DP ("cells counts the cells actually evaluated and the full-matrix figure
                         # is reported alongside"): the figures an unbanded pass over the same alignments has
                         "full_matrix": {"algorithmic_bytes_per_launch": alg_bytes - st["dp_bytes"] + st["dp_bytes_full"],
                                         "achieved": (alg_bytes - st["dp_bytes"] + st["dp_bytes_full"]) / step_s / 1e9,
                                         "gcups": st["dp_cells_full"] / step_s / 1e9},
                         "banded_alignments": st["n_banded"], "code_wave_alignments": st["n_code_wave"], "band_redone": st["n_band_redone"], "band_redo_why": st["band_redo_why"],
                         "phase_clocks": st["phase_clocks"], "work_groups_per_cu": st["wg_per_cu"],
                         # the small-window kernel (one wave per window, graph in LDS): windows it polished / sent back to
                         # poa_window_kernel2 (the retry pass is inside step_kernel_ms) and why
                         "small_windows": st["n_small"], "small_bailed": st["n_small_bailed"], "small_bail_why": st["small_bail_why"],
                         "small_work": dict(zip(("alignments", "dp_rows", "subgraph_chunks", "traceback_boxes", "traceback_regathers", "subgraph_intervals"), st["small_work"]))},
        }
        if world > 1:
            # A 1 -> N curve drawn from `value` alone would mix workloads: N = 1 is cfg2 (2000 windows, resident all at once: the batch
            # ends with its deepest window's chain), N > 1 is cfg3 / N per rank (12 500 ... 50 000 windows: the long-queue rate).  What the
            # N-GPU number has to be read against is THE SAME JOB on one GPU (cfg3 whole: `bench.py --config cfg3`), printed here.
            out["per_gpu_value"] = out["value"] / world
            n1 = {"workload": "cfg3 whole (50 Mbp, 100 000 windows) on ONE GPU: `python bench.py --config cfg3`", "measured_in_this_run": False}
            for f in ("profiles/r06/bench_cfg3_1gpu.json", "profiles/r05/bench_cfg3_1gpu.json", "profiles/r03/h_bench_cfg3_1gpu.json"):
                try:
                    j1 = json.load(open(os.path.join(ROOT, f)))
                    n1.update({"value": j1["value"], "ms_per_step": j1["ms_per_step"], "source": f})
                    break
                except Exception:
                    continue
            if scaling == "strong" and "value" in n1:
                n1["speedup_over_same_job_on_one_gpu"] = out["value"] / n1["value"]
                n1["efficiency_vs_same_job"] = out["value"] / n1["value"] / world
            out["n1_same_job"] = n1
            out["scaling_note"] = ("N > 1 polishes cfg3 (50 Mbp / N per rank); the driver's N = 1 line is cfg2 (1 Mbp). value(N) / value(1) therefore mixes "
                                   "a batch-size effect into the curve: use n1_same_job for the one-GPU reference of THIS job")
        elif not a.no_rccl_selftest and not a.no_upload_leg and a.dist_backend == "nccl":
            out["rccl_selftest"] = rccl_selftest(eng, batch, local_rank)
        if do_product:
            # the product on the same workload (same seed -> the same windows, tests/test_synth_files.py)
            try:
                th = max(1, min(32, len(os.sched_getaffinity(0))))
                out["product_polish"] = {}
                for name, paths in pfiles:
                    same_windows = name in ("cfg2", "cfg3")             # (the packed batch of this run holds the same windows)
                    out["product_polish"][name] = product_polish(paths, pwindow, (m, x, g), th, expect=b"".join(res.consensus) if same_windows else None,
                                                                 reps=1 if name == "cfg3" else 2, batches=a.product_batches)
                    if name != "cfg3":
                        # like for like with the reference's GPU path, which packs and copies its windows inside polish(): host-built windows
                        hb = product_polish(paths, pwindow, (m, x, g), th, expect=b"".join(res.consensus) if same_windows else None, reps=2,
                                            batches=a.product_batches, mode="0")
                        out["product_polish"][name]["host_built_in_process"] = {k: hb[k] for k in ("polish_s", "windows_per_s", "windows_per_s_incl_initialize", "runs", "fasta_matches_kernel_leg")}
                    out["product_polish"][name]["cli"] = product_cli(paths, pwindow, (m, x, g), th, expect=b"".join(res.consensus) if same_windows else None,
                                                                     reps=1 if name == "cfg3" else 2, batches=a.product_batches)
                    if a.no_device_modes or name == "cfg3":
                        continue
                    # SURVEY 8(f) rows 1, 2, 4 on the interval they were built for (reference: CUDAPolisher::find_overlap_breaking_points
                    # sits inside the same timed program, src/cuda/cudapolisher.cpp:74-214):
                    #   (cli, above)   SAM, the binary's default: CIGAR walk + window construction in HBM at the end of initialize() (rcn_engine_build_windows_from_cigars)
                    #   host_built     SAM, RACON_HIP_DEVICE_WINDOWS=0: Window::add_layer on the host, chunks packed and streamed inside polish()
                    #   device_align   PAF, --cudaaligner-batches 1: pairwise alignment + CIGAR walk + construction in HBM (rcn_engine_build_windows_from_pairs)
                    #   host_align     PAF, the default: the host's edlib-equivalent inside initialize(), then the host-built path
                    # (PAF and SAM inputs differ in their alignments, hence in their windows: the PAF legs are compared with each other)
                    ex = b"".join(res.consensus) if same_windows else None
                    # (the binary's default is `auto`: construction in HBM when it fits -- the "cli" leg above; here the host-built path,
                    #  windows packed per chunk inside polish(), RACON_HIP_DEVICE_WINDOWS=0)
                    dm = {"host_built": product_cli(paths, pwindow, (m, x, g), th, expect=ex, reps=2, batches=a.product_batches, env_add={"RACON_HIP_DEVICE_WINDOWS": "0"})}
                    if "error" not in dm["host_built"] and "error" not in out["product_polish"][name]["cli"]:
                        dm["host_built"]["fasta_matches_device_built"] = dm["host_built"]["fasta_md5"] == out["product_polish"][name]["cli"]["fasta_md5"]
                    if os.path.exists(paths.get("paf", "")):
                        dm["device_align"] = product_cli(paths, pwindow, (m, x, g), th, reps=2, batches=a.product_batches, overlaps="paf", flags=("--cudaaligner-batches", "1"))
                        dm["host_align"] = product_cli(paths, pwindow, (m, x, g), th, reps=1, batches=a.product_batches, overlaps="paf")
                        if "error" not in dm["device_align"] and "error" not in dm["host_align"]:
                            dm["device_align"]["fasta_matches_host_aligner"] = dm["device_align"]["fasta_md5"] == dm["host_align"]["fasta_md5"]
                        if dm["device_align"].get("wall_s") and dm["host_align"].get("wall_s"):
                            dm["device_align"]["whole_binary_speedup_over_host_aligner"] = dm["host_align"]["wall_s"] / dm["device_align"]["wall_s"]
                    out["product_polish"][name]["device_modes"] = dm
                out["value_product_polish"] = out["product_polish"][pfiles[0][0]]["windows_per_s"]
                if "cfg3_share" in out["product_polish"]:
                    out["value_product_polish_12k"] = out["product_polish"]["cfg3_share"]["windows_per_s"]
                # ... and through the drop-in binary (the Logger's own line)
                for name, key in ((pfiles[0][0], "value_product_polish_cli"), ("cfg3_share", "value_product_polish_12k_cli")):
                    cli = out["product_polish"].get(name, {}).get("cli", {})
                    if "windows_per_s" in cli:
                        out[key] = cli["windows_per_s"]
            except Exception as e:                       # the headline line must not die with the product leg
                out["product_polish"] = {"error": repr(e)}
        if pmulti is not None:
            # N > 1: the product is ONE process on all N devices (the other ranks idle at the barrier below)
            try:
                th = max(1, min(32 * world, len(os.sched_getaffinity(0))))
                out["product_multi_device"] = product_multi_device(pmulti, a.window, (m, x, g), th, world)
                out["value_product_polish"] = out["product_multi_device"]["all_devices"]["windows_per_s"]
            except Exception as e:
                out["product_multi_device"] = {"error": repr(e)}
        if not a.no_cpu and world == 1:
            # CPU baseline on this box's host cores (rank 0 at N = 1 only): the oracle's AVX2 int16 variant (the scheme of spoa's SIMD
            # engine: row vectors + log-step prefix max), whole batch; the thread count is swept and the box's best is what the
            # GPU is compared with.  The scalar int32 oracle is timed next to it on a sample for reference.
            from oracle import oracle_lib
            lim = cpu_limits()
            ncpu = lim["sched_affinity"] or (os.cpu_count() or 1)
            # bounded sample of the same workload: at most 2000 windows (all of cfg2)
            cb = batch if batch.n_windows <= 2000 else batch.select(range(2000))
            want = [int(v) for v in a.cpu_threads.split(",") if v] or [8, 16, 24, 32, 48, 64, 96, 128, ncpu]
            sweep_threads = sorted({min(max(1, v), ncpu) for v in want})
            oracle_lib.consensus(cb.select(range(min(64, cb.n_windows))), m, x, g, True, min(ncpu, 32), simd=True)   # warm up
            sweep, best_dt, ref, cores = {}, None, None, ncpu
            for th in sweep_threads:
                bt = None
                for _ in range(2):
                    tc = time.perf_counter()
                    ref = oracle_lib.consensus(cb, m, x, g, True, th, simd=True)
                    dtc = time.perf_counter() - tc
                    bt = dtc if bt is None else min(bt, dtc)
                sweep[str(th)] = cb.n_windows / bt
                if best_dt is None or bt < best_dt:
                    best_dt, cores = bt, th
            ok = ref.consensus == res.consensus[:cb.n_windows]
            n_s = a.cpu_sample or min(cb.n_windows, max(64, 4 * min(ncpu, 64)))
            sample = cb.select(range(n_s))
            tc = time.perf_counter()
            oracle_lib.consensus(sample, m, x, g, True, min(ncpu, cores))
            dts = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": cb.n_windows / best_dt, "unit": "windows/s", "cores": cores, "kind": "port",
                                   "sample": "%d windows of the same workload, oracle/poa_oracle.cpp AVX2 int16 variant, best thread "
                                             "count of the sweep (%d; the process may run on %d logical CPUs, cgroup quota %s), best of 2 (%.2f s); "
                                             "scalar int32 oracle on the first %d windows, %d threads: %.0f windows/s"
                                             % (cb.n_windows, cores, ncpu, lim.get("cgroup_cpus"), best_dt, n_s, min(ncpu, cores), n_s / dts),
                                   "thread_sweep_windows_per_s": sweep, "host": lim,
                                   "matches_gpu": bool(ok)}
            # what "all host cores, same box" would be: the per-thread rate from the part of the sweep the lease can actually run
            # (threads <= the cgroup quota), times the box's physical cores -- an extrapolation, said as such
            quota = lim.get("cgroup_cpus") or ncpu
            under = {int(t): v for t, v in sweep.items() if int(t) <= quota + 0.5} or {min(int(t) for t in sweep): sweep[str(min(int(t) for t in sweep))]}
            t_q = max(under)
            per_thread = under[t_q] / t_q
            phys = physical_cores()
            cb_ = out["cpu_baseline"]
            cb_["per_thread_windows_per_s"] = per_thread
            cb_["per_thread_from"] = "%d threads (<= the %.0f-CPU quota): %.0f windows/s" % (t_q, quota, under[t_q])
            cb_["physical_cores"] = phys
            cb_["extrapolated_all_cores"] = per_thread * phys if phys else None
            cb_["note"] = ("value = measured on the CPUs this process may use (cgroup quota %s of %d logical CPUs); extrapolated_all_cores = "
                           "per_thread_windows_per_s x physical_cores, what the whole box would give at perfect scaling (not measured)" % (lim.get("cgroup_cpus"), ncpu))
        # THE METRIC AS SURVEY.md 8(d) DEFINES IT, first class: windows/s on the polish() interval of the product (files ->
        # Polisher; in-process and through the binary), next to the resident-input kernel leg (`value`), each against the CPU
        # number of this line
        cpu_v = out.get("cpu_baseline", {}).get("value")
        kernel_leg = out["value"]
        out["value_kernel_leg"] = kernel_leg
        out["value_is"] = "kernel leg: the timed loop of this line (inputs resident in HBM), windows x steps / (steps x ms_per_step)"
        if out.get("value_product_polish"):
            first = out.get("product_polish", {}).get(pfiles[0][0], {}) if pfiles else {}
            out["product"] = {"definition": "polished windows/s on the Polisher::polish() interval (reference src/polisher.cpp:493 -> :539-543)",
                              "value": out["value_product_polish"], "value_cli": out.get("value_product_polish_cli"), "unit": "windows/s",
                              "workload": pfiles[0][0] if pfiles else ("%d devices, one process" % world),
                              "fraction_of_kernel_leg": out["value_product_polish"] / kernel_leg if world == 1 else None,
                              # the same interval when the windows are packed and uploaded INSIDE it (the reference's GPU path does that:
                              # src/cuda/cudapolisher.cpp:254-276), and with initialize() counted in
                              "value_host_built": (first.get("host_built_in_process") or {}).get("windows_per_s"),
                              "value_incl_initialize": first.get("windows_per_s_incl_initialize"),
                              "vs_cpu_baseline": (out["value_product_polish"] / cpu_v) if cpu_v else None,
                              "vs_cpu_baseline_cli": (out["value_product_polish_cli"] / cpu_v) if (cpu_v and out.get("value_product_polish_cli")) else None}
            if world == 1 and not a.config and contig == 1_000_000:
                # THE HEADLINE IS THE CONTRACT METRIC: windows/s on the product's polish() interval, in-process, on the
                # configuration the metric is quoted on.  The kernel leg stays in the line as value_kernel_leg; `ms_per_step`, `roofline` and
                # `steps` describe the timed kernel loop (value_kernel_leg = windows x steps / (steps x ms_per_step)).
                out["value"] = out["value_product_polish"]
                out["value_is"] = ("product: polished windows/s on the Polisher::polish() interval (in-process, files -> racon_amd/host Polisher, windows built in HBM "
                                   "at the end of initialize()); value_kernel_leg = the timed loop of this line (ms_per_step, roofline)")
        if cpu_v:
            out["cpu_baseline"]["gpu_kernel_leg_over_cpu"] = kernel_leg / cpu_v
            allc = out["cpu_baseline"].get("extrapolated_all_cores")
            if allc:
                out["cpu_baseline"]["gpu_kernel_leg_over_cpu_all_cores_extrapolated"] = kernel_leg / allc
                if "product" in out:
                    out["product"]["vs_cpu_baseline_all_cores_extrapolated"] = out["product"]["value"] / allc
        if a.verify:
            from oracle import oracle_lib
            ref = oracle_lib.consensus(batch, m, x, g, True, 0, simd=True)
            out["verified_windows"] = int(sum(ref.consensus[i] == res.consensus[i] for i in range(batch.n_windows)))
            out["verified_flags"] = bool((ref.polished == res.polished).all() and (ref.chimeric == res.chimeric).all())
        print(json.dumps(out), flush=True)
    if world > 1:
        # The other ranks wait for rank 0's product leg on the HOST (a key in the process group's store), not inside a collective: an
        # NCCL barrier entered early is a kernel that spins on every other GPU until rank 0 arrives -- on exactly the devices the
        # product leg's one racon_hip process is polishing on (and a launch that wants whole compute units to itself could wait for
        # that kernel for ever).
        try:
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("racon_bench_rank0_done", "1")
            else:
                store.wait(["racon_bench_rank0_done"], datetime.timedelta(minutes=90))
        except Exception:
            pass
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
