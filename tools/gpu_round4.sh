#!/bin/bash
# Round-4 GPU visits.  Usage: bash tools/gpu_round4.sh <tag> [what...]
#   what: small (the small-window kernel's tests) cfg4 (bench line of BASELINE configs[3]) tests (whole GPU suite) bench
#         w1000 prof (rocprofv3 kernel stats of the bench) pmc (HBM traffic passes) sq (issue counters) prof4 (cfg4 profile build)
set -u
TAG=${1:-r04a}; shift || true
WHAT=${*:-small cfg4}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export RCN_EXPERIMENT=1
has() { [[ " $WHAT " == *" $1 "* ]]; }
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -12) > "$OUT/box.txt" 2>&1
benchline() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%s: %.0f windows/s  step %.2f ms  frac %.3f  gcups %.0f  small %s bailed %s why %s  phase %s' % ('$1', j['value'], r['step_kernel_ms'], r['frac'], r.get('gcups',0), r.get('small_windows'), r.get('small_bailed'), r.get('small_bail_why'), [round(c/1e9,1) for c in r.get('phase_clocks',[])]))"; }

if has small; then
  timeout 1500 python -m pytest tests/test_gpu_small.py -m gpu -q -x --durations=8 > "$OUT/pytest_small.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_small.log"; tail -25 "$OUT/pytest_small.log"
fi
if has cfg4; then
  for k in 1 2; do
    timeout 900 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2> "$OUT/bench_cfg4.err" | tee "$OUT/bench_cfg4_$k.json" | benchline "cfg4 run $k"
  done
  RCN_NO_SMALL=1 timeout 900 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2>> "$OUT/bench_cfg4.err" | tee "$OUT/bench_cfg4_nosmall.json" | benchline "cfg4 without the small kernel"
  tail -3 "$OUT/bench_cfg4.err"
fi
if has percu; then
  for n in 8 12 16 20 24; do
    RCN_SMALL_PER_CU=$n timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2>/dev/null | benchline "cfg4, $n windows per CU"
  done | tee "$OUT/cfg4_per_cu.txt"
fi
if has tests; then
  timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -30 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 1500 python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if has w1000; then
  timeout 900 python bench.py --config w1000 --steps 5 --warmup 1 --no-product --no-upload-leg 2> "$OUT/bench_w1000.err" | tee "$OUT/bench_w1000.json" | benchline w1000
fi
