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
