"""Summarises rocprofv3 --pmc counter_collection CSVs under <out>/pmc_<COUNTER>/ :
per kernel, per counter: dispatches, sum, mean per dispatch.  Units are left
raw here; the gfx950 corrections of MI355X_MICROARCH.md (HBM section) are applied
by whoever quotes the numbers (DESIGN.md / bench.py --traffic)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
import json
per_kernel = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [set(), 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
                acc[k][0].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
                acc[k][1] += float(row.get("Counter_Value", 0) or 0)
        for (kn, cn), (ids, total) in sorted(acc.items()):
            n = max(1, len(ids))
            print("%s | %s | dispatches %d | sum %.6g | per dispatch %.6g" % (kn, cn, n, total, total / n))
            if "poa_window_kernel" in kn:            # (the consensus kernels: poa_window_kernel2, its _deep instance, poa_window_kernel_small)
                tot = per_kernel.setdefault("rcn::poa_window_kernel*", {}).setdefault(cn, [0.0, 0])
                tot[0] += total; tot[1] += n
# traffic.json: HBM bytes per launch of the consensus kernel.  Raw units are KB (rocprofv3 derived counters);
# FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM) prescribes for gfx950 wide reads, WRITE_SIZE is uncalibrated.
# A step of bench.py is one launch of the kernel, or -- split launch -- two concurrent ones on disjoint CU sets: the bench line
# of the counter pass says which (roofline.launches_per_step); the traffic quoted next to the roofline is per STEP.
lps = 1
for name in ("pmc_FETCH_SIZE.json", "prof_bench.json"):
    try:
        line = open(os.path.join(out, name)).read().strip().splitlines()[-1]
        lps = int(json.loads(line)["roofline"].get("launches_per_step", 1))
        break
    except Exception:
        pass
for kn, c in per_kernel.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # per step: all dispatches of the kernel's instances / steps, steps = dispatches / launches per step
        fetch = c["FETCH_SIZE"][0] / max(1, c["FETCH_SIZE"][1]) * 1024.0 * 2.0 * lps
        write = c["WRITE_SIZE"][0] / max(1, c["WRITE_SIZE"][1]) * 1024.0 * lps
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from srchash import kernel_source_hash
        import time
        json.dump({"kernel": kn, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                   "bytes_per_launch": fetch + write, "launches_per_step": lps, "kernel_source_hash": kernel_source_hash(),
                   "measured": "%s, %s" % (os.path.basename(os.path.normpath(out)), time.strftime("%Y-%m-%d")),
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB -> bytes, FETCH_SIZE x2 (gfx950 correction); per bench step "
                           "= per dispatch x launches_per_step (the split launch is two concurrent dispatches of the kernel)"},
                  open(os.path.join(out, "traffic.json"), "w"))
        print("traffic.json:", fetch + write, "bytes per launch")
