#!/bin/bash
# Round-3 GPU visit (see tools/gpu_round.sh for the profile passes): parity tests, the bench line with the product legs
# and the CPU sweep, the split-launch A/B, the host-core probe.
# Usage: bash tools/gpu_round3.sh <tag> [what...]   what: tests bench split probe debug prof pmc big
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
set -u
TAG=${1:-r03a}; shift || true
WHAT=${*:-tests bench split probe debug}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
(nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -20) > "$OUT/box.txt" 2>&1

if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
  tail -30 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 1200 python bench.py --steps 5 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if has split; then
  B="python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
  run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.0f windows/s  step %.2f ms  launches %s  frac %.3f  split %s' % (j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac'], r['split_launch']))"; }
  { run "no split" RCN_SPLIT=0
    run "default (96 CUs, 1 per CU)" RCN_X=0
    run "64 CUs x1" RCN_SPLIT_CUS=64
    run "128 CUs x1" RCN_SPLIT_CUS=128
    run "64 CUs x2" RCN_SPLIT_CUS=64 RCN_SPLIT_DEEP_PER_CU=2
    run "96 CUs x2" RCN_SPLIT_CUS=96 RCN_SPLIT_DEEP_PER_CU=2
    run "128 CUs x2" RCN_SPLIT_CUS=128 RCN_SPLIT_DEEP_PER_CU=2
    run "96 CUs x1, rest 6 per CU" RCN_SPLIT_REST_PER_CU=6
    run "no split again" RCN_SPLIT=0
    run "default again" RCN_X=0
  } > "$OUT/split_ab.txt" 2>&1
  cat "$OUT/split_ab.txt"
fi
if has probe; then
  timeout 300 python tools/cpu_probe.py > "$OUT/cpu_probe.json" 2>&1; cat "$OUT/cpu_probe.json"
fi
if has tests2; then
  timeout 2400 python -m pytest tests/test_gpu_product_path.py tests/test_gpu_unbounded_shapes.py tests/test_gpu_band.py tests/test_cli_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bench_contract.py -m gpu -q -x --durations=8 > "$OUT/pytest_gpu2.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu2.log"
  tail -40 "$OUT/pytest_gpu2.log"
fi
if has timeline; then
  # host timelines of the product's polish(): cfg2 (one chunk) and one GPU's share of cfg3 (12 500 windows, four chunks)
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
for c, s in ((1_000_000, 20260921), (6_250_000, 20260922)):
    print(bench.product_files(c, 30.0, s, 32))
PY
  for F in /tmp/racon_amd_cache/files_1000000_30_20260921 /tmp/racon_amd_cache/files_6250000_30_20260922; do
    for k in 1 2 3; do
      RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/timeline_$(basename $F)_$k.err" | md5sum
    done
    grep -E "racon::|polish:|piece|collect|reserve" "$OUT/timeline_$(basename $F)_3.err" | head -80
  done
fi
if has batches; then
  # the product legs with one and with two batch objects per device (-c 1 / -c 2: two / four engines)
  for c in 1 2; do
    timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu --no-upload-leg --product-batches $c 2> "$OUT/bench_c$c.err" | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j.get('product_polish',{})
print('-c $c:', {k:(round(v['polish_s']*1e3,1), round(v['windows_per_s'])) if isinstance(v,dict) and 'polish_s' in v else v for k,v in p.items()}, 'value', round(j['value']))"
  done 2>&1 | tee "$OUT/product_batches.txt"
fi
if has chunksweep; then
  # engines per device x chunk size on one GPU's share of cfg3 (12 500 windows): the polish() interval of the binary
  F=/tmp/racon_amd_cache/files_6250000_30_20260922
  [ -d "$F" ] || python -c "
import os,sys; sys.path.insert(0, os.getcwd()); import bench; bench.product_files(6_250_000, 30.0, 20260922, 32)"
  { for E in 2 3 4; do for CW in 0 1600 2100 3200 4200 6300; do
      best=999
      for k in 1 2; do
        t=$(RACON_HIP_ENGINES_PER_DEVICE=$E RACON_HIP_CHUNK_WINDOWS=$CW racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2>&1 >/dev/null | grep "generated consensus" | sed 's/.*consensus \([0-9.]*\) s/\1/')
        best=$(python -c "print(min($best, float('$t')))")
      done
      echo "engines $E chunk $CW: polish $best s"
    done; done; } 2>&1 | tee "$OUT/chunksweep.txt"
fi
if has malloc; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/malloc_time tools/probe/malloc_time.hip && /tmp/malloc_time > "$OUT/malloc_time.txt" 2>&1; cat "$OUT/malloc_time.txt"
  /tmp/malloc_time > "$OUT/malloc_time_second_process.txt" 2>&1; head -3 "$OUT/malloc_time_second_process.txt"
fi
if has debug; then
  # host timeline of one product polish() (RCN_DEBUG prints the engine's own clock)
  F=/tmp/racon_amd_cache/files_1000000_30_20260921
  if [ -d "$F" ]; then
    for k in 1 2; do RCN_DEBUG=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/debug_polish_$k.err" | md5sum; done
    grep -E "racon::|polish:|piece" "$OUT/debug_polish_2.err" | head -40
    RACON_HIP_NO_WARMUP=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/debug_polish_nowarm.err" | md5sum
    grep -E "racon::" "$OUT/debug_polish_nowarm.err"
  fi
fi
if has big; then
  # cfg3 at full size on one GPU (100 000 windows through the engine's queue) with every window checked against the oracle,
  # and one GPU's share of cfg5 (scale 0.125 = 250 k windows)
  cat /sys/fs/cgroup/memory.max /proc/meminfo 2>/dev/null | head -4; df -h /tmp | tail -1
  timeout 2400 python bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu --verify > "$OUT/bench_cfg3_1gpu.json" 2> "$OUT/bench_cfg3_1gpu.err"
  echo "cfg3 exit $?"; cut -c1-3000 "$OUT/bench_cfg3_1gpu.json"; tail -3 "$OUT/bench_cfg3_1gpu.err"
fi
if has cfg5; then
  timeout 2400 python tools/cfg5_at_size.py --scale ${CFG5_SCALE:-0.125} > "$OUT/cfg5_at_size.json" 2> "$OUT/cfg5_at_size.err"
  echo "cfg5 exit $?"; cat "$OUT/cfg5_at_size.json"; tail -5 "$OUT/cfg5_at_size.err"
fi
if has winprof; then
  # per-window phase clocks (RCN_PROF_WIN build): where the deepest windows of the bench batch spend their time
  make -s -C racon_amd/csrc prof
  for C in "" "--config cfg4"; do
    T=$(echo "$C" | tr -d ' -' ); T=${T:-cfg2}
    RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so timeout 600 python bench.py $C --steps 1 --warmup 1 --no-cpu --no-product --no-upload-leg > "$OUT/winprof_${T}_bench.json" 2> "$OUT/winprof_$T.txt"
    echo "== winprof $T"; grep -v amdgpu.ids "$OUT/winprof_$T.txt" | tail -12
  done
fi
if has others; then
  # the other workloads of BASELINE.json through the same kernel (bench lines with roofline + cpu_baseline)
  for C in cfg4 w1000 cfg5x0.004; do
    timeout 900 python bench.py --config $C --steps 5 --warmup 1 > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.err"
    echo "== $C exit $?"; python - "$OUT/bench_$C.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]; c = j.get("cpu_baseline") or {}
print("%s | %.0f windows/s, step %.2f ms, frac %.3f, cpu %.0f (%s threads) -> %.1fx" % (j["config"]["workload"][:70], j["value"], r["step_kernel_ms"], r["frac"], c.get("value", 0), c.get("cores"), j["value"] / max(1, c.get("value", 1))))
PY
  done
  timeout 900 python bench.py --contig 4000000 --steps 3 --warmup 1 --no-cpu --no-product > "$OUT/bench_4mbp.json" 2> "$OUT/bench_4mbp.err"; python -c "
import json; j=json.loads(open('$OUT/bench_4mbp.json').read().strip().splitlines()[-1]); print('4 Mbp: %.0f windows/s frac %.3f' % (j['value'], j['roofline']['frac']))"
fi
BENCH_PROF="python bench.py --steps 3 --warmup 1 --no-cpu --no-upload-leg --no-product"
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
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh "$OUT"
