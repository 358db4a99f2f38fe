#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats + HBM counters.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh [tag] [what...]
#   what: tests bench prof pmc build winprof   (default: tests bench prof pmc build)
# Everything lands under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
set -u
TAG=${1:-r01}; shift || true
WHAT=${*:-tests bench prof pmc build}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }

(nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; rocm-smi --showproductname 2>/dev/null | head -20) > "$OUT/box.txt" 2>&1

if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 900 python bench.py --steps 5 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
  # the CPU baseline's thread-count sweep of that run, as a table
  python - "$OUT/bench.json" > "$OUT/cpu_scaling.txt" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = j["cpu_baseline"]
print("# CPU baseline (oracle/poa_oracle.cpp, AVX2 int16 variant), same windows as the GPU line, best of 3 per thread count")
print("# threads  windows/s")
for th, v in sorted(c["thread_sweep_windows_per_s"].items(), key=lambda kv: int(kv[0])):
    print("%7s  %9.1f%s" % (th, v, "   <- best (cpu_baseline.value)" if int(th) == c["cores"] else ""))
print("# GPU value %.1f windows/s = %.2fx ; incl. upload %.1f windows/s" % (j["value"], j["value"] / c["value"], j["value_incl_upload"]))
PY
  cat "$OUT/cpu_scaling.txt"
fi
if has winprof; then
  # per-window phase clocks (RCN_PROF_WIN build): where the deepest windows of the bench batch spend their time
  make -s -C racon_amd/csrc prof
  RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu > "$OUT/winprof_bench.json" 2> "$OUT/winprof.txt"
  grep -A6 "per-window clocks" "$OUT/winprof.txt" | tail -8
fi
BENCH_PROF="python bench.py --steps 3 --warmup 1 --no-cpu --no-upload-leg"
if has prof; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- $BENCH_PROF > "$OUT/prof_bench.json" 2> "$OUT/prof.err"
  echo "prof exit $?"; cat "$OUT/prof_bench.json"
  find "$OUT/prof" -name "*kernel_stats.csv" -exec head -5 {} \;
fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH_PROF > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err"
    echo "pmc $C exit $?"
  done
  python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1; cat "$OUT/pmc_summary.txt"
fi
if has build; then
  # window construction in HBM (rcn_engine_build_windows): timings + rocprofv3 kernel stats of the same command
  timeout 600 python tools/build_bench.py --oracle > "$OUT/build_bench.json" 2> "$OUT/build_bench.err"
  echo "build exit $?"; cat "$OUT/build_bench.json"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/build_prof" -o trace -- python tools/build_bench.py --contig 8000000 --reps 3 > "$OUT/build_bench_8m.json" 2> "$OUT/build_prof.err"
  cat "$OUT/build_bench_8m.json"
  find "$OUT/build_prof" -name "*kernel_stats.csv" -exec sh -c 'cut -c1-160 "$1" | head -8' _ {} \;
fi
# keep the merge-back small: drop bulky raw traces, keep stats + counter tables
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh "$OUT"
