#!/bin/bash
# GPU visit: whole GPU test tier, bench line, aligner bench.
set -u
TAG=${1:-r02f}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rs > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -8 "$OUT/pytest_gpu.log"
RCN_DEBUG=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu > "$OUT/bench_dbg.json" 2> "$OUT/bench_dbg.err"; grep "polish:\|streamed" "$OUT/bench_dbg.err" | tail -6
python - "$OUT/bench_dbg.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f incl upload %.0f (%.1f%% lower) launch %.2f ms step %.2f / %.2f ms" % (j["value"], j["value_incl_upload"], 100 * (1 - j["value_incl_upload"] / j["value"]), j["roofline"]["avg_launch_ms"], j["ms_per_step"], j["ms_per_step_incl_upload"]))
PY
timeout 600 python tools/align_bench.py > "$OUT/align_bench.json" 2> "$OUT/align_bench.err"; echo "align bench exit $?"; cat "$OUT/align_bench.json"; tail -3 "$OUT/align_bench.err"
