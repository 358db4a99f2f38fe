"""Occupancy of the device by the consensus kernels over a product run, from a rocprofv3 --kernel-trace CSV:
for the interval first kernel start .. last kernel end, how long 0 / 1 / 2 / >= 3 consensus kernels were in flight,
and every kernel's interval relative to the first start.  usage: kernel_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "poa_window_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("rcn::", ""),
                             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
rows.sort()
if not rows:
    sys.exit("no consensus kernels in the trace")
t0 = rows[0][0]
t1 = max(r[1] for r in rows)
ev = sorted([(a, 1) for a, _, _, _ in rows] + [(b, -1) for _, b, _, _ in rows])
busy = {}
lvl, last = 0, t0
for t, d in ev:
    busy[min(lvl, 3)] = busy.get(min(lvl, 3), 0) + (t - last)
    lvl += d
    last = t
span = t1 - t0
print("consensus kernels: %d, first start .. last end %.2f ms" % (len(rows), span / 1e6))
for k in sorted(busy):
    print("  %s in flight: %6.2f ms (%4.1f %%)" % (("%d" % k) if k < 3 else ">= 3", busy[k] / 1e6, 100.0 * busy[k] / span))
for a, b, n, wg in rows:
    print("  %-28s work-groups %5d  %7.2f .. %7.2f ms  (%6.2f ms)" % (n, wg, (a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6))
