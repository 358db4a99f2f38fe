#!/bin/bash
# GPU visit: device pairwise aligner bring-up + streamed-upload timing.
set -u
TAG=${1:-r02e}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_pair_align.py -q -x --timeout 300 > "$OUT/pair_tests.log" 2>&1; echo "pair tests exit $?" >> "$OUT/pair_tests.log"; tail -30 "$OUT/pair_tests.log"
RCN_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu > "$OUT/bench_dbg.json" 2> "$OUT/bench_dbg.err"; grep "polish:\|streamed" "$OUT/bench_dbg.err" | tail -12
python - "$OUT/bench_dbg.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f incl upload %.0f (%.1f%% lower) launch %.2f ms step %.2f / %.2f ms" % (j["value"], j["value_incl_upload"], 100 * (1 - j["value_incl_upload"] / j["value"]), j["roofline"]["avg_launch_ms"], j["ms_per_step"], j["ms_per_step_incl_upload"]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_window_build.py -q -x --timeout 600 > "$OUT/other_tests.log" 2>&1; echo "exit $?" >> "$OUT/other_tests.log"; tail -4 "$OUT/other_tests.log"
