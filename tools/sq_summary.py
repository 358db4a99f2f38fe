"""Summarises the rocprofv3 --pmc passes of SQ counters made by tools/gpu_round*.sh (sq_<label>_<n>/): per consensus kernel,
counter sums per dispatch, and the derived secondary ceilings SURVEY.md 8(d) asks for next to the HBM roofline:

  valu_issue_frac = cycles the SIMDs' vector ALUs were issuing / (dispatch duration x the SIMDs the dispatch ran on)
  lds_frac        = cycles the LDS arrays were busy / (dispatch duration x the CUs the dispatch ran on)

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_*
count quad-cycles (x 4 = shader cycles); SQ_INSTS_* count wave-instructions; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count
LDS-array cycles.

What a dispatch's DURATION and its SHARE OF THE CHIP are (round 6: the r04 / r05 files got both wrong for the split launch):
  * duration: from the kernel trace of THE SAME pass (rocprofv3 --kernel-trace next to --pmc: End - Start per dispatch) when the pass has
    one; otherwise from the pass's own bench line (HIP events).  rocprofv3 --pmc SERIALISES the dispatches of a process: the two
    "concurrent" launches of the split then run one after the other and the second launch's HIP-event interval CONTAINS the first
    (launch_ms = [16.1, 33.4], step 33.4).  Detected (step ~ max(launch_ms) and the two launches cannot have overlapped) and undone:
    the second launch's own duration is launch_ms[1] - launch_ms[0].
  * share of the chip: the split launch runs under CU masks -- the deep instance on `deep_cus` CUs, the other one on the rest
    (roofline.split_launch of the bench line) -- so the counters of each are normalised by its own CUs x 4 SIMDs, not by 1024.
Writes <out>/issue_<label>.json (stamped with the kernel-source hash, like traffic.json)."""
import csv, glob, json, os, sys, time
from collections import defaultdict

CLOCK_GHZ, N_CU = 2.4, 256


def kernel_short(name):
    return name.split("(")[0].replace("rcn::", "")


def pass_durations(d, bench_line):
    """{kernel: (duration ms of one dispatch, source)} for one counter pass."""
    out = {}
    # 1. the kernel trace of the same pass
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        acc = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                kn = row.get("Kernel_Name", "?")
                if "poa_window_kernel" not in kn:
                    continue
                try:
                    acc[kernel_short(kn)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
                except (KeyError, ValueError):
                    pass
        for kn, v in acc.items():
            out[kn] = (sum(v) / len(v), "kernel trace of the pass (%d dispatches)" % len(v))
    if out or not bench_line:
        return out
    # 2. the bench line of the pass (HIP events)
    r = bench_line["roofline"]
    step, lm = r["step_kernel_ms"], r.get("launch_ms") or []
    if r.get("launches_per_step", 1) == 2 and len(lm) == 2:
        deep, other, src = lm[0], lm[1], "HIP events of the pass"
        # serialised by the profiler: the second launch's interval contains the first (it was enqueued at the same time and waited)
        if lm[1] > 1.5 * lm[0] and abs(step - lm[1]) < 0.05 * step:
            other, src = lm[1] - lm[0], "HIP events of the pass, serialised by the profiler: launch_ms[1] - launch_ms[0]"
        out["poa_window_kernel2_deep"] = (deep, src)
        out["poa_window_kernel2"] = (other, src)
    else:
        out["*"] = (step, "HIP events of the pass")
    return out


def chip_share(kn, bench_line):
    """(CUs, SIMDs) a dispatch of this kernel ran on."""
    sl = ((bench_line or {}).get("roofline") or {}).get("split_launch") or {}
    deep_cus = int(sl.get("deep_cus") or 0)
    if deep_cus:
        cus = deep_cus if "deep" in kn else N_CU - deep_cus - int(sl.get("mid_cus") or 0)
        return cus, 4 * cus
    return N_CU, 4 * N_CU


def summarise(out, label, log=print):
    acc = defaultdict(lambda: defaultdict(lambda: [set(), 0.0]))
    durs = defaultdict(list)            # kernel -> [(ms, source)] over the passes
    bench_line = None
    for d in sorted(glob.glob(os.path.join(out, "sq_%s_*" % label))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    kn = row.get("Kernel_Name", "?")
                    if "poa_window_kernel" not in kn:
                        continue
                    a = acc[kernel_short(kn)][row.get("Counter_Name", "?")]
                    a[0].add(row.get("Dispatch_Id", "")); a[1] += float(row.get("Counter_Value", 0) or 0)
        line = None
        try:
            line = json.loads(open(d + ".json").read().strip().splitlines()[-1])
            bench_line = line
        except Exception:
            pass
        for kn, v in pass_durations(d, line).items():
            durs[kn].append(v)
    res = {}
    for kn, cs in sorted(acc.items()):
        per = {c: v[1] / max(1, len(v[0])) for c, v in cs.items()}
        nd = max(len(v[0]) for v in cs.values())
        log("== %s (%d dispatches)" % (kn, nd))
        for c in sorted(per):
            log("   %-24s %14.6g per dispatch" % (c, per[c]))
        dv = durs.get(kn) or durs.get("*")
        if not dv:
            continue
        dur = sum(v[0] for v in dv) / len(dv)
        n_cu, n_simd = chip_share(kn, bench_line)
        cyc = dur * 1e-3 * CLOCK_GHZ * 1e9
        d = {"dispatch_ms": dur, "dispatch_ms_source": dv[0][1], "cus": n_cu, "simds": n_simd}
        if "SQ_ACTIVE_INST_VALU" in per: d["valu_issue_frac"] = per["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * n_simd)
        if "SQ_ACTIVE_INST_SCA" in per: d["scalar_issue_frac"] = per["SQ_ACTIVE_INST_SCA"] * 4 / (cyc * n_simd)
        if "SQ_LDS_IDX_ACTIVE" in per: d["lds_frac"] = per["SQ_LDS_IDX_ACTIVE"] / (cyc * n_cu)
        if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_share"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
        if "SQ_WAVE_CYCLES" in per:
            wc = per["SQ_WAVE_CYCLES"]
            for k, nm in (("SQ_ACTIVE_INST_ANY", "wave_time_issuing"), ("SQ_WAIT_ANY", "wave_time_parked_waitcnt"), ("SQ_WAIT_INST_ANY", "wave_time_issue_stalled")):
                if k in per: d[nm] = per[k] / wc
            d["waves_resident_avg"] = wc * 4 / cyc
            d["waves_resident_avg_per_simd"] = wc * 4 / cyc / n_simd
        tot = sum(per.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        if tot:
            d["wave_instructions"] = tot
            d["cycles_per_instruction_per_simd"] = cyc * n_simd / tot
            d["valu_share_of_instructions"] = per.get("SQ_INSTS_VALU", 0) / tot
        res[kn] = d
        log("   -> " + ", ".join("%s %s" % (k, ("%.4g" % v) if isinstance(v, float) else v) for k, v in d.items()))
    return res


def main():
    out, label = sys.argv[1], sys.argv[2]
    res = summarise(out, label)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from srchash import kernel_source_hash
    json.dump({"label": label, "kernels": res, "kernel_source_hash": kernel_source_hash(), "measured": "%s, %s" % (os.path.basename(os.path.normpath(out)), time.strftime("%Y-%m-%d")),
               "note": "rocprofv3 --pmc SQ counters (three passes of eight), quad-cycle counters x 4; per kernel: duration of ITS dispatch (kernel trace of the pass, "
                       "or HIP events with the profiler's serialisation undone) x the SIMDs / CUs of ITS CU mask"},
              open(os.path.join(out, "issue_%s.json" % label), "w"), indent=1)


if __name__ == "__main__":
    main()
