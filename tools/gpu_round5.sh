#!/bin/bash
# Round-5 GPU visits.  Usage: bash tools/gpu_round5.sh <tag> [what...]
#   what: subset (this round's new / changed tests) tests (whole GPU suite) bench (default line) fuzz (tools/fuzz_sweep.py)
#         cfg4 w1000 (bench lines of the other workloads) final (kernel stats + HBM / SQ counter passes + all lines)
#         tiers (the three-tier split sweep) cfg4prod (short-read product, chunk plans) cfg5 (cfg5 whole on one GPU) cfg3bin (cfg3 whole through the binary)
set -u
TAG=${1:-r05a}; shift || true
WHAT=${*:-subset bench}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export RCN_EXPERIMENT=1
has() { [[ " $WHAT " == *" $1 "* ]]; }
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -12) > "$OUT/box.txt" 2>&1
benchline() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
sl=r.get('split_launch') or {}
print('%s: %.0f windows/s  step %.2f ms  launches %s mid %s  frac %.3f  gcups %.0f  small %s bailed %s why %s' % ('$1', j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], sl.get('mid_launch_ms'), r['frac'], r.get('gcups',0), r.get('small_windows'), r.get('small_bailed'), r.get('small_bail_why')))"; }
QB="--steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"

if has subset2; then
  timeout 2400 python -m pytest tests/test_gpu_small.py tests/test_cli_e2e.py tests/test_gpu_product_path.py tests/test_gpu_window_build.py -m gpu -q --durations=10 > "$OUT/pytest_subset2.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_subset2.log"; tail -25 "$OUT/pytest_subset2.log"
fi
if has subset; then
  timeout 2400 python -m pytest tests/test_gpu_small.py tests/test_cli_e2e.py tests/test_gpu_bench_contract.py \
      "tests/test_gpu_product_path.py::test_polish_interval_chunks_and_warmup" "tests/test_gpu_fullsize.py::test_one_rank_rccl_exchange" \
      -m gpu -q -x --durations=10 > "$OUT/pytest_subset.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_subset.log"; tail -25 "$OUT/pytest_subset.log"
fi
if has tests; then
  timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -30 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 1500 python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; cut -c1-3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if has fuzz; then
  timeout 1500 python tools/fuzz_sweep.py --seeds ${FUZZ_SEEDS:-120} --out "$OUT/fuzz_sweep.json" > "$OUT/fuzz_sweep.log" 2>&1
  echo "fuzz exit $?"; tail -4 "$OUT/fuzz_sweep.log" | cut -c1-1500
fi
if has fuzz2; then
  # a second, disjoint set of seeds on the closing tree
  timeout 1500 python tools/fuzz_sweep.py --first-seed 20000 --seeds ${FUZZ2_SEEDS:-200} --out "$OUT/fuzz_sweep_second_set.json" > "$OUT/fuzz_sweep_second_set.log" 2>&1
  echo "fuzz2 exit $?"; tail -1 "$OUT/fuzz_sweep_second_set.log" | cut -c1-900
  timeout 900 python tools/fuzz_sweep.py --no-small --first-seed 30000 --seeds ${FUZZ2_K2_SEEDS:-80} --out "$OUT/fuzz_sweep_kernel2_second_set.json" > "$OUT/fuzz_sweep_kernel2_second_set.log" 2>&1
  echo "fuzz2 (kernel2) exit $?"; tail -1 "$OUT/fuzz_sweep_kernel2_second_set.log" | cut -c1-700
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
if has fuzzk2; then
  timeout 1500 python tools/fuzz_sweep.py --no-small --first-seed 9000 --seeds ${FUZZ_K2_SEEDS:-60} --out "$OUT/fuzz_sweep_kernel2.json" > "$OUT/fuzz_sweep_kernel2.log" 2>&1
  echo "fuzz (kernel2) exit $?"; tail -2 "$OUT/fuzz_sweep_kernel2.log" | cut -c1-1200
fi
if has cfg4; then
  for k in 1 2; do
    timeout 900 python bench.py --config cfg4 $QB 2> "$OUT/bench_cfg4.err" | tee "$OUT/bench_cfg4_$k.json" | benchline "cfg4 run $k"
  done
fi
if has w1000; then
  timeout 900 python bench.py --config w1000 --steps 5 --warmup 1 --no-product --no-upload-leg --no-cpu 2> "$OUT/bench_w1000.err" | tee "$OUT/bench_w1000.json" | benchline w1000
fi
if has tiers; then
  # the split launch's plan: deep tier (one work-group per CU, the _deep instance), middle tier (RCN_SPLIT_MID windows at RCN_SPLIT_MID_PER_CU
  # per CU on RCN_SPLIT_MID_CUS CUs), rest -- cfg2, ms per step and per launch
  run() { echo "== $1"; shift; env "$@" python bench.py $QB 2>/dev/null | benchline "$*"; }
  {
    run "default"
    run "default again"
    for spec in ${TIER_SPECS:-"64:4:16" "96:4:24" "128:4:32" "96:2:48" "64:2:32" "160:4:40" "128:6:24"}; do
      IFS=: read n per cus <<< "$spec"
      run "mid $n windows, $per per CU, $cus CUs" RCN_SPLIT_MID=$n RCN_SPLIT_MID_PER_CU=$per RCN_SPLIT_MID_CUS=$cus
    done
    run "default, third time"
  } 2>&1 | tee "$OUT/tiers.txt"
fi
if has repro; then
  timeout 900 python tools/exp/fuzz_repro.py ${REPRO_ARGS:-5088 115 1,-1,-1 1} > "$OUT/fuzz_repro.txt" 2>&1
  echo "repro exit $?"; cut -c1-400 "$OUT/fuzz_repro.txt" | head -150
fi
if has timeline2; then
  # polish() of the binary on cfg2-shaped files, by mode: host-built windows (0), CIGAR walk + construction in HBM (2, SAM),
  # alignment + walk + construction in HBM (--cudaaligner-batches 1, PAF) -- host timeline of the last of three runs each
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(1_000_000, 30.0, 20260921, 32))
PY
  F=/tmp/racon_amd_cache/files_1000000_30_20260921
  for spec in "mode0:sam:" "mode2:sam:RACON_HIP_DEVICE_WINDOWS=2" "mode3:paf:RACON_HIP_DEVICE_WINDOWS=3" "mode2late:sam:RACON_HIP_DEVICE_WINDOWS=2 RACON_HIP_BUILD_IN_POLISH=1"; do
    IFS=: read name ovl envs <<< "$spec"
    for k in 1 2 3; do
      env $envs RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.$ovl $F/targets.fasta 2> "$OUT/timeline_cfg2_$name.err" | md5sum | cut -c1-8
      echo "cfg2 $name: $(grep 'generated consensus' $OUT/timeline_cfg2_$name.err) | $(grep 'total =' $OUT/timeline_cfg2_$name.err)"
    done
    grep -E "racon::|polish:|piece|collect|reserve|pass of|timing|pairs:" "$OUT/timeline_cfg2_$name.err" | cut -c1-300 > "$OUT/timeline_cfg2_$name.txt"
  done 2>&1 | tee "$OUT/timeline2.txt"
fi
if has shardtime; then
  # where a shard's time goes when a job is cut into more window ranges than devices (cfg5 at a quarter of its size, three sequential shards)
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from racon_amd.synth import simulate_fragment_files
SC = float(os.environ.get("SHARD_SCALE", "0.25")); d = "/tmp/racon_amd_cache/cfg5_%g" % SC
if not os.path.exists(d + "/.done"):
    p = simulate_fragment_files(d, int(33_333_333 * SC), int(100_000 * SC), seed=20260924); open(d + "/.done", "w").write(str(p["n_overlaps"]))
PY
  F=/tmp/racon_amd_cache/cfg5_${SHARD_SCALE:-0.25}
  RACON_HIP_DEVICE_SHARDS=${SHARD_N:-3} RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -f -t 32 --cudaaligner-batches 1 $F/reads.fastq $F/overlaps.paf $F/reads.fastq 2> "$OUT/shardtime.err" | md5sum
  grep -E "racon::|polish:|piece|collect|reserve|pass of|timing|pairs:|racon_hip" "$OUT/shardtime.err" | cut -c1-260 > "$OUT/shardtime.txt"; tail -60 "$OUT/shardtime.txt"
fi
if has inittime; then
  # initialize() with device-side construction: where its time goes at 12 500 and 50 000 windows (uploads out of pageable memory staged through pinned slots)
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(6_250_000, 30.0, 20260922, 32))
print(bench.product_files(25_000_000, 30.0, 20260922, 32))
PY
  for F in /tmp/racon_amd_cache/files_6250000_30_20260922 /tmp/racon_amd_cache/files_25000000_30_20260922; do
    for e in "" "RACON_HIP_DEVICE_WINDOWS=0"; do
      for k in 1 2; do
        env $e RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/inittime.err" | md5sum | cut -c1-8
        echo "$(basename $F) [$e]: $(grep -E 'transformed data|generated consensus|total =|built on device' $OUT/inittime.err | sed 's/.racon::Polisher:://' | tr '\n' '|' | cut -c1-600)"
      done
    done
  done 2>&1 | tee "$OUT/inittime.txt"
fi
if has variants2; then
  # more scheduling variants of engine_deep.hip (libracon_hip_v<k>.so, built by hand: see profiles/r05/i_deep_tu_variants.txt), A/B on one box
  for k in 1 2; do
    python bench.py $QB 2>/dev/null | benchline "as shipped ($k)"
    for v in ${VARIANTS:-v1 v3 v5}; do [ -f racon_amd/csrc/libracon_hip_$v.so ] && RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_$v.so python bench.py $QB 2>/dev/null | benchline "libracon_hip_$v.so ($k)"; done
  done 2>&1 | tee "$OUT/variants2.txt"
fi
if has variants; then
  # compiler-scheduling variants of the consensus kernels (built by hand into libracon_hip_exp*.so: -mllvm -amdgpu-sched-strategy=max-ilp on
  # engine_deep.hip alone / on engine.hip and engine_deep.hip), A/B on one box: cfg2, and 8000 windows for the long-queue rate
  for k in 1 2; do
    python bench.py $QB 2>/dev/null | benchline "as shipped ($k)"
    for v in exp exp2; do [ -f racon_amd/csrc/libracon_hip_$v.so ] && RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_$v.so python bench.py $QB 2>/dev/null | benchline "libracon_hip_$v.so ($k)"; done
  done 2>&1 | tee "$OUT/variants.txt"
  python bench.py --contig 4000000 $QB 2>/dev/null | benchline "4 Mbp as shipped" | tee -a "$OUT/variants.txt"
  RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_exp2.so python bench.py --contig 4000000 $QB 2>/dev/null | benchline "4 Mbp libracon_hip_exp2.so" | tee -a "$OUT/variants.txt"
fi
if has sleep; then
  # code waves polling less often (libracon_hip_exp.so: -DRCN_HELP_SLEEP=8), A/B/A/B on one box
  for k in 1 2; do
    python bench.py $QB 2>/dev/null | benchline "as shipped ($k)"
    RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_exp.so python bench.py $QB 2>/dev/null | benchline "experiment build ($k)"
  done 2>&1 | tee "$OUT/exp_build.txt"
fi
if has build; then
  # window construction in HBM, with the CIGAR walk (32-bit positions since round 5): kernel times on cfg2's shape and on 8 Mbp; its tests
  timeout 900 python -m pytest tests/test_gpu_window_build.py -m gpu -q > "$OUT/pytest_window_build.log" 2>&1; tail -3 "$OUT/pytest_window_build.log"
  for c in 1000000 8000000; do timeout 600 python tools/build_bench.py --contig $c 2>/dev/null | tail -1 | tee "$OUT/build_bench_$c.json" | cut -c1-700; done
fi
if has splitcus; then
  # CUs (= windows, one per CU) of the deep launch, once more on this round's kernels
  run() { echo "== $1"; shift; env "$@" python bench.py $QB 2>/dev/null | benchline "$*"; }
  { for c in 16 24 32 64; do run "deep launch on $c CUs" RCN_SPLIT_CUS=$c; done; run "deep launch on 32 CUs, 48 windows in it (two rounds for 16 CUs)" RCN_SPLIT_CUS=32 RCN_SPLIT_DEEP=48; } 2>&1 | tee "$OUT/splitcus.txt"
fi
if has cfg4prod; then
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(1_000_000, 60.0, 20260923, 32, short_reads=True))
PY
  F=/tmp/racon_amd_cache/files_1000000_60_20260923_short
  for e in "" "RACON_HIP_DEVICE_WINDOWS=0" "RACON_HIP_DEVICE_WINDOWS=0 RACON_HIP_CHUNK_WINDOWS=5000" ${CFG4PROD_EXTRA:-}; do
    for k in 1 2 3; do
      env $e RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 -w 200 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/cfg4prod.err" | md5sum | cut -c1-8
      echo "cfg4 files [$e]: $(grep 'generated consensus' $OUT/cfg4prod.err) $(grep -c 'engine .* chunk' $OUT/cfg4prod.err) chunks"
    done
  done | tee "$OUT/cfg4prod.txt"
  RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 -w 200 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/timeline_cfg4.err" | md5sum
  grep -E "racon::|polish:|piece|collect|reserve|pass of|timing" "$OUT/timeline_cfg4.err" | head -80 > "$OUT/timeline_cfg4_product.txt"
fi
# ---- issue / LDS counters (SURVEY 8(d) secondary ceilings): rocprofv3 --pmc passes of SQ counters, 8 per pass ----
sqpasses() {   # $1 = label, rest = bench arguments
  local label=$1; shift
  local n=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
    n=$((n + 1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d "$OUT/sq_${label}_$n" -o pmc -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu --no-product --no-upload-leg > "$OUT/sq_${label}_$n.json" 2> "$OUT/sq_${label}_$n.err"
    echo "sq $label pass $n exit $?"; tail -2 "$OUT/sq_${label}_$n.err" | cut -c1-200
  done
  python tools/sq_summary.py "$OUT" "$label" > "$OUT/sq_${label}_summary.txt" 2>&1; cat "$OUT/sq_${label}_summary.txt"
}
if has final; then
  BP="--steps 3 --warmup 1 --no-cpu --no-upload-leg --no-product"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python bench.py $BP > "$OUT/prof_bench.json" 2> "$OUT/prof.err"
  find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \; ; head -6 "$OUT/kernel_stats.csv" | cut -c1-200
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof4" -o trace -- python bench.py --config cfg4 $BP > "$OUT/prof_bench_cfg4.json" 2> "$OUT/prof4.err"
  find "$OUT/prof4" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_cfg4.csv" \; ; head -5 "$OUT/kernel_stats_cfg4.csv" | cut -c1-200
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python bench.py $BP > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err"; echo "pmc $C exit $?"
  done
  python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1; tail -3 "$OUT/pmc_summary.txt"
  mkdir -p "$OUT/c4"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/c4/pmc_$C" -o pmc -- python bench.py --config cfg4 $BP > "$OUT/c4/pmc_$C.json" 2> "$OUT/c4/pmc_$C.err"; echo "pmc cfg4 $C exit $?"
  done
  python tools/pmc_summary.py "$OUT/c4" > "$OUT/pmc_summary_cfg4.txt" 2>&1; tail -3 "$OUT/pmc_summary_cfg4.txt"
  sqpasses cfg4 --config cfg4
  sqpasses cfg2
  timeout 900 python bench.py --config cfg4 --steps 10 --warmup 2 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4_full.err"; benchline cfg4 < "$OUT/bench_cfg4.json"
  timeout 900 python bench.py --config w1000 --steps 5 --warmup 1 --no-product --no-upload-leg 2> "$OUT/bench_w1000.err" | tee "$OUT/bench_w1000.json" | benchline w1000
  timeout 900 python bench.py --config cfg5x0.004 --steps 5 --warmup 1 --no-product --no-upload-leg 2> "$OUT/bench_cfg5.err" | tee "$OUT/bench_cfg5x0.004.json" | benchline cfg5x0.004
  timeout 900 python bench.py --contig 4000000 --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2> "$OUT/bench_4mbp.err" | tee "$OUT/bench_4mbp.json" | benchline 4mbp
fi
if has sq2; then sqpasses cfg2; fi
if has sq4; then sqpasses cfg4 --config cfg4; fi
if has cfg5; then
  timeout ${CFG5_TIMEOUT:-2400} python tools/cfg5_whole.py ${CFG5_ARGS:---scale 1.0 --shards 8} > "$OUT/cfg5_whole.json" 2> "$OUT/cfg5_whole.err"
  echo "cfg5 exit $?"; tail -3 "$OUT/cfg5_whole.err" | cut -c1-400; cut -c1-3500 "$OUT/cfg5_whole.json"
fi
if has cfg3bin; then
  timeout 2400 python bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu --no-upload-leg --product-contig 0 > "$OUT/bench_cfg3_1gpu.json" 2> "$OUT/bench_cfg3.err"
  echo "cfg3 exit $?"; cut -c1-2500 "$OUT/bench_cfg3_1gpu.json"; tail -3 "$OUT/bench_cfg3.err"
fi
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT" | tail -1
