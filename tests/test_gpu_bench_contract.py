"""-m gpu: bench.py prints ONE JSON line with the contract's keys (driver contract + the `roofline` / `cpu_baseline` objects),
single rank and two ranks on one GPU (torch.distributed with gloo, the code path the driver launches with torchrun)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _line(out: bytes) -> dict:
    # exactly one JSON object on stdout (with --dist-backend gloo the gloo library also chats there, possibly mid-line)
    text = out.decode()
    assert text.count('{"metric"') == 1, text
    obj, _ = json.JSONDecoder().raw_decode(text[text.index('{"metric"'):])
    return obj


def test_bench_single_rank_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--contig", "100000", "--cpu-sample", "64"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT).stdout
    j = _line(out)
    assert KEYS <= set(j), KEYS - set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["unit"] == "windows/s" and j["dtype"] == "int16" and j["scaling"] == "weak" and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["achieved"] > 0 and abs(j["value"] - 200 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    assert j["value_kernel_leg"] == j["value"] and j["value_is"].startswith("kernel leg")      # (no product leg on this workload)
    # the step is one launch or two concurrent ones (split launch): the interval they cover is what the bytes are divided by
    assert r["launches_per_step"] == len(r["launch_ms"]) in (1, 2) and r["step_kernel_ms"] >= max(r["launch_ms"]) * 0.999
    assert r["step_kernel_ms"] <= j["ms_per_step"] * 1.001 and abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["step_kernel_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "windows/s" and c["sample"]
    assert c["matches_gpu"] is True and max(c["thread_sweep_windows_per_s"].values()) == c["value"]
    assert c["host"]["sched_affinity"] >= 1 and "cgroup_cpus" in c["host"]          # what "all host cores" is on this box
    # ... and what the whole box would be worth: per-thread rate from the part of the sweep the lease can run x physical cores
    assert c["per_thread_windows_per_s"] > 0 and c["physical_cores"] >= 1
    assert abs(c["extrapolated_all_cores"] - c["per_thread_windows_per_s"] * c["physical_cores"]) < 1e-6 * c["extrapolated_all_cores"]
    assert abs(c["gpu_kernel_leg_over_cpu_all_cores_extrapolated"] - j["value_kernel_leg"] / c["extrapolated_all_cores"]) < 1e-9 * j["value"]
    # the RCCL code path of the multi-GPU line, exercised on a one-rank group (untimed)
    assert j["rccl_selftest"]["ok"] is True and j["rccl_selftest"]["backend"] == "nccl", j["rccl_selftest"]
    # the upload-inclusive rate (pack + H2D + kernel + D2H per step) is reported next to `value`, never instead of it
    assert 0 < j["value_incl_upload"] <= j["value"] * 1.05 and j["ms_per_step_incl_upload"] > 0


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--contig", "100000", "--no-cpu", "--dist-backend", "gloo", "--product-multi-contig", "300000"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=env).stdout
    j = _line(out)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 200 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6      # whole-job aggregate over both ranks
    # N > 1 also carries the PRODUCT: one racon_hip process on every visible device against the same files on one device
    pm = j["product_multi_device"]
    assert pm["devices"] == 2 and pm["fasta_identical"] is True and pm["all_devices"]["windows"] == 600, pm
    assert j["product"]["value"] == pm["all_devices"]["windows_per_s"]
    # the curve cannot read as super-linear: per-GPU value and the one-GPU reference of the SAME job are in the line
    assert abs(j["per_gpu_value"] - j["value"] / 2) < 1e-9 * j["value"] and "n1_same_job" in j and "scaling_note" in j
    assert j["n1_same_job"]["measured_in_this_run"] is False and j["n1_same_job"].get("value", 0) > 0


def test_gpus_flag_spawns_its_ranks_or_refuses():
    """`python bench.py --gpus 2` without a launcher starts its own two ranks; a launcher that started another number of ranks
    than --gpus says is an error, not a line for the wrong N."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--contig", "100000",
                          "--no-cpu", "--no-product", "--dist-backend", "gloo"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}).stdout
    assert _line(out)["n_gpus"] == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--contig", "100000", "--no-cpu", "--no-product"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and b"refusing" in bad.stderr and b'{"metric"' not in bad.stdout


def test_product_on_eight_logical_devices():
    """The N > 1 product leg before an 8-GPU node ever sees it: one racon_hip process over RACON_HIP_FAKE_DEVICES=8 (eight logical
    devices on this one GPU: racon_amd/host/polisher.cpp drives each with its own engines off one cursor, reference
    src/cuda/cudapolisher.cpp:228-240, 254-276) prints the FASTA of the one-device run."""
    sys.path.insert(0, ROOT)
    import bench
    files = bench.product_files(400_000, 30.0, 20260977, 8)
    r = bench.product_multi_device(files, 500, (3, -5, -4), 16, 8, fake=True)
    assert "error" not in r, r
    assert r["fasta_identical"] is True and r["all_devices"]["windows"] == r["one_device"]["windows"] == 800


def test_bench_product_legs_on_the_default_workload():
    """The default line (cfg2) also carries the PRODUCT: the same workload as files through racon_amd/host's Polisher -- the
    Logger-bracketed polish() interval, in-process and through the binary -- with the FASTA checked against the kernel leg."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--product-contig", "0"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT).stdout
    j = _line(out)
    p = j["product_polish"]["cfg2"]
    assert p["windows"] == 2000 and p["fasta_matches_kernel_leg"] is True and 0 < p["polish_s"] < 1.0
    assert p["cli"]["windows"] == 2000 and p["cli"]["fasta_matches_kernel_leg"] is True and 0 < p["cli"]["polish_s"] < 1.0
    assert abs(j["value_product_polish"] - 2000 / p["polish_s"]) < 1e-6 * j["value_product_polish"]
    # ... first class: the metric on the interval SURVEY.md 8(d) defines it on, next to the kernel leg
    assert j["product"]["value"] == j["value_product_polish"] and j["product"]["value_cli"] == p["cli"]["windows_per_s"]
    assert 0 < j["product"]["fraction_of_kernel_leg"] <= 1.05
    # the headline IS the contract metric on this workload; the kernel leg (what ms_per_step and the roofline describe) is next to it
    assert j["value"] == j["product"]["value"] and j["value_is"].startswith("product")
    assert abs(j["value_kernel_leg"] - 2000 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value_kernel_leg"]
    assert abs(j["product"]["fraction_of_kernel_leg"] - j["value"] / j["value_kernel_leg"]) < 1e-9
    # ... and the like-for-like interval (windows packed and uploaded inside polish()) plus initialize()-inclusive rate
    hb = p["host_built_in_process"]
    assert hb["fasta_matches_kernel_leg"] is True and j["product"]["value_host_built"] == hb["windows_per_s"] > 0
    assert 0 < j["product"]["value_incl_initialize"] < j["product"]["value"]
    assert j["config"]["windows_per_gpu"] == 2000 and "cfg2" in j["config"]["workload"]
    # SURVEY 8(f) rows on the interval they were built for: CIGAR walk + construction in HBM (SAM), alignment + walk + construction
    # in HBM (PAF, --cudaaligner-batches 1) against the host aligner on the same PAF
    dm = p["device_modes"]
    assert dm["host_built"]["fasta_matches_kernel_leg"] is True and dm["host_built"]["fasta_matches_device_built"] is True
    assert dm["device_align"]["fasta_matches_host_aligner"] is True and dm["device_align"]["windows"] == dm["host_align"]["windows"] == 2000
    # (typically 3-4x faster; a generous bound, not a race, on a loaded box)
    assert dm["device_align"]["wall_s"] < 2.0 * dm["host_align"]["wall_s"]
