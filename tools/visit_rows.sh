#!/bin/bash
# GPU visit: per-row-class clocks of the banded DP; cfg4 / cfg5 bench lines with per-window phase clocks.
set -u
TAG=${1:-r02g}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
L=$PWD/racon_amd/csrc
RACON_HIP_LIB=$L/libracon_hip_rows.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu > "$OUT/rows_bench.json" 2> "$OUT/rows.txt"; grep -A10 "banded DP rows" "$OUT/rows.txt" | tail -11
for CFG in cfg4 cfg5x0.004 w1000; do
  echo "== $CFG"
  timeout 600 python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  python - "$OUT/bench_$CFG.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]; pc = r["phase_clocks"]; tot = float(sum(pc)) or 1.0
print("%s | %.0f w/s (incl. upload %.0f) launch %.2f ms frac %.3f gcups %.0f | banded %s redone %s | sub %.1f desc %.1f dp %.1f tb %.1f add %.1f merge %.1f cons %.1f" % (
    j["config"]["workload"][:60], j["value"], j["value_incl_upload"], r["avg_launch_ms"], r["frac"], r["gcups"], r["banded_alignments"], r["band_redone"], *[100 * v / tot for v in pc[:7]]))
PY
  RACON_HIP_LIB=$L/libracon_hip_prof.so timeout 600 python bench.py --config $CFG --steps 1 --warmup 1 --no-cpu > /dev/null 2> "$OUT/winprof_$CFG.txt"; grep -A4 "per-window clocks" "$OUT/winprof_$CFG.txt" | tail -5
done
