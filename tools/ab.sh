#!/bin/bash
# A/B of environment-switched kernel variants on one box: tools/ab.sh <tag> <reps> "<envA>" "<envB>" ...
# Alternates the variants (launch times differ by a few % between processes), prints kernel ms / windows/s per run.
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
set -u
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for rep in $(seq 1 "$REPS"); do
  k=0
  for V in "$@"; do
    k=$((k+1))
    env $V timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu ${AB_ARGS:-} > "$OUT/ab_${k}_$rep.json" 2> "$OUT/ab_${k}_$rep.err" || echo "variant $k failed"
    python - "$OUT/ab_${k}_$rep.json" "$V" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    pc = r.get("phase_clocks") or [0] * 8
    tot = float(sum(pc)) or 1.0
    print("%-34s %8.1f w/s (incl. upload %8.1f)  launch %.2f ms  banded %s redone %s why %s | phases %% sub %.1f desc %.1f dp %.1f tb %.1f add %.1f merge %.1f cons %.1f other %.1f | Gclk %.1f" % (
        sys.argv[2] or "(default)", j["value"], j.get("value_incl_upload") or 0.0, r["avg_launch_ms"], r.get("banded_alignments"), r.get("band_redone"), r.get("band_redo_why"),
        *[100.0 * v / tot for v in pc], tot / 1e9))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
  done
done
