"""Summarises the rocprofv3 --pmc passes of SQ counters made by tools/gpu_round4.sh (sq_<label>_<n>/): per consensus kernel,
counter sums per dispatch, and the derived secondary ceilings SURVEY.md 8(d) asks for next to the HBM roofline:

  valu_issue_frac = cycles the SIMDs' vector ALUs were issuing / (dispatch duration x SIMDs)
  lds_frac        = cycles the LDS arrays were busy / (dispatch duration x CUs)

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_*
count quad-cycles (x 4 = shader cycles); SQ_INSTS_* count wave-instructions; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count
LDS-array cycles.  The dispatch duration comes from the bench line of the same pass (roofline.step_kernel_ms: HIP events).
Writes <out>/issue_<label>.json (stamped with the kernel-source hash, like traffic.json)."""
import csv, glob, json, os, sys, time
from collections import defaultdict

out, label = sys.argv[1], sys.argv[2]
CLOCK_GHZ, N_CU, N_SIMD = 2.4, 256, 1024
acc = defaultdict(lambda: defaultdict(lambda: [set(), 0.0]))
step_ms, launches = None, 1
for d in sorted(glob.glob(os.path.join(out, "sq_%s_*" % label))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                kn = row.get("Kernel_Name", "?")
                if "poa_window_kernel" not in kn:
                    continue
                kn = kn.split("(")[0].replace("rcn::", "")
                a = acc[kn][row.get("Counter_Name", "?")]
                a[0].add(row.get("Dispatch_Id", "")); a[1] += float(row.get("Counter_Value", 0) or 0)
    try:
        j = json.loads(open(d + ".json").read().strip().splitlines()[-1])
        step_ms = j["roofline"]["step_kernel_ms"]; launches = j["roofline"].get("launches_per_step", 1)
        launch_ms = j["roofline"].get("launch_ms")
    except Exception:
        pass
res = {}
for kn, cs in sorted(acc.items()):
    per = {c: v[1] / max(1, len(v[0])) for c, v in cs.items()}
    nd = max(len(v[0]) for v in cs.values())
    print("== %s (%d dispatches)" % (kn, nd))
    for c in sorted(per):
        print("   %-24s %14.6g per dispatch" % (c, per[c]))
    if step_ms:
        dur = step_ms
        if launches == 2 and launch_ms:          # split launch: the deep instance is launch 0, the other one launch 1
            dur = launch_ms[0] if "deep" in kn else launch_ms[1]
        cyc = dur * 1e-3 * CLOCK_GHZ * 1e9
        d = {"dispatch_ms": dur}
        if "SQ_ACTIVE_INST_VALU" in per: d["valu_issue_frac"] = per["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * N_SIMD)
        if "SQ_ACTIVE_INST_SCA" in per: d["scalar_issue_frac"] = per["SQ_ACTIVE_INST_SCA"] * 4 / (cyc * N_SIMD)
        if "SQ_LDS_IDX_ACTIVE" in per: d["lds_frac"] = per["SQ_LDS_IDX_ACTIVE"] / (cyc * N_CU)
        if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_share"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
        if "SQ_WAVE_CYCLES" in per:
            wc = per["SQ_WAVE_CYCLES"]
            for k, nm in (("SQ_ACTIVE_INST_ANY", "wave_time_issuing"), ("SQ_WAIT_ANY", "wave_time_parked_waitcnt"), ("SQ_WAIT_INST_ANY", "wave_time_issue_stalled")):
                if k in per: d[nm] = per[k] / wc
            d["waves_resident_avg"] = wc * 4 / cyc
        tot = sum(per.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        if tot:
            d["wave_instructions"] = tot
            d["cycles_per_instruction_per_simd"] = cyc * N_SIMD / tot
            d["valu_share_of_instructions"] = per.get("SQ_INSTS_VALU", 0) / tot
        res[kn] = d
        print("   -> " + ", ".join("%s %.4g" % kv for kv in d.items()))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from srchash import kernel_source_hash
json.dump({"label": label, "kernels": res, "kernel_source_hash": kernel_source_hash(), "measured": "%s, %s" % (os.path.basename(os.path.normpath(out)), time.strftime("%Y-%m-%d")),
           "note": "rocprofv3 --pmc SQ counters (three passes of eight), quad-cycle counters x 4; duration = HIP-event time of the dispatch in the same pass"},
          open(os.path.join(out, "issue_%s.json" % label), "w"), indent=1)
