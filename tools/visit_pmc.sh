#!/bin/bash
# GPU visit: SQ counters for banded vs unbanded rows, streamed-upload timeline, cfg4 phase profile.
set -u
OUT=gpurun_out/${1:-r02d}; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_band.py -q -x --timeout 600 > "$OUT/band_tests.log" 2>&1; echo "band tests exit $?" >> "$OUT/band_tests.log"; tail -4 "$OUT/band_tests.log"
# streamed upload timeline
RCN_DEBUG=1 timeout 300 python - > "$OUT/stream_debug.txt" 2>&1 <<'PY'
import time
from racon_amd.engine import HipEngine
from racon_amd.synth import config_windows
b = config_windows("cfg2")
eng = HipEngine(3, -5, -4, True)
for k in range(3):
    t = time.perf_counter(); eng.consensus(b); dt = time.perf_counter() - t
    st = eng.stats()
    print("streamed call %d: wall %.2f ms kernel span %.2f ms h2d %.2f d2h %.2f launches %d" % (k, dt * 1e3, st["kernel_ms"], st["h2d_ms"], st["d2h_ms"], st["n_launches"]), flush=True)
eng.upload(b)
for k in range(2):
    t = time.perf_counter(); eng.run_only(); dt = time.perf_counter() - t
    print("resident run %d: wall %.2f ms kernel %.2f" % (k, dt * 1e3, eng.stats()["kernel_ms"]), flush=True)
t = time.perf_counter(); eng.upload(b); print("plain upload wall %.2f ms" % ((time.perf_counter() - t) * 1e3))
PY
cat "$OUT/stream_debug.txt" | grep -v amdgpu.ids
bash tools/ab.sh ${1:-r02d} 2 "RCN_NO_BAND=1" "RCN_X=0"
AB_ARGS="--config cfg4" bash tools/ab.sh ${1:-r02d}_cfg4 1 "RCN_X=0"
BENCH_PROF="python bench.py --steps 3 --warmup 1 --no-cpu"
for V in noband band; do
  if [ $V = noband ]; then export RCN_NO_BAND=1; else unset RCN_NO_BAND; fi
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmcsq_$V" -o pmc -- $BENCH_PROF > "$OUT/pmcsq_$V.json" 2> "$OUT/pmcsq_$V.err"
  echo "pmc sq $V exit $?"
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmcsq2_$V" -o pmc -- $BENCH_PROF > "$OUT/pmcsq2_$V.json" 2> "$OUT/pmcsq2_$V.err"
  echo "pmc sq2 $V exit $?"
done
unset RCN_NO_BAND
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmcsq*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [set(), 0.0])
        for row in csv.DictReader(open(f)):
            if "poa_window_kernel2" not in row.get("Kernel_Name", ""): continue
            k = row.get("Counter_Name", "?")
            acc[k][0].add(row.get("Dispatch_Id", "")); acc[k][1] += float(row.get("Counter_Value", 0) or 0)
        print(os.path.basename(d), " ".join("%s=%.4g" % (k, v[1] / max(1, len(v[0]))) for k, v in sorted(acc.items())))
PY
