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
print('%s: %.0f windows/s  step %.2f ms  frac %.3f  gcups %.0f  small %s bailed %s why %s work %s phase %s' % ('$1', j['value'], r['step_kernel_ms'], r['frac'], r.get('gcups',0), r.get('small_windows'), r.get('small_bailed'), r.get('small_bail_why'), r.get('small_work'), [round(c/1e9,1) for c in r.get('phase_clocks',[])]))"; }

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
# ---- issue / LDS counters (SURVEY 8(d) secondary ceilings): rocprofv3 --pmc passes of SQ counters, 8 per pass ----
# usage: ... sq           -> cfg4 (small-window kernel) and cfg2 (poa_window_kernel2 + the deep instance)
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
if has sq; then
  rocprofv3 -L > "$OUT/counters_list.txt" 2>&1; grep -c "SQ_" "$OUT/counters_list.txt"
  sqpasses cfg4 --config cfg4
  sqpasses cfg2
fi
if has sq4; then sqpasses cfg4 --config cfg4; fi
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT" | tail -1
if has align; then
  timeout 600 python tools/align_bench.py > "$OUT/align_bench.json" 2> "$OUT/align_bench.err"; echo "align exit $?"; tail -3 "$OUT/align_bench.json" | cut -c1-600
fi
if has testsq; then   # quick subset: the suites touched this round
  timeout 1500 python -m pytest tests/test_gpu_small.py tests/test_gpu_pair_align.py tests/test_gpu_bench_contract.py tests/test_cli_e2e.py -m gpu -q -x --durations=8 > "$OUT/pytest_subset.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_subset.log"; tail -15 "$OUT/pytest_subset.log"
fi
# ---- the round's closing set: kernel stats + HBM counters of the default line and of cfg4, the other workloads ----
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
  find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null; find "$OUT" -name "*.db" -delete 2>/dev/null
fi
if has sq8k; then sqpasses 4mbp --contig 4000000; fi
if has timeline4; then
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(1_000_000, 60.0, 20260923, 32, short_reads=True))
PY
  F=/tmp/racon_amd_cache/files_1000000_60_20260923_short
  for k in 1 2 3; do
    RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 -w 200 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/timeline_cfg4_$k.err" | md5sum
  done
  grep -E "racon::|polish:|piece|collect|reserve|pass of" "$OUT/timeline_cfg4_3.err" | head -60
fi
if has hwq; then
  # HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): two engines x five streams share them, and a
  # copy stream that lands behind the other engine's running kernel waits for it.  polish() of the binary on one GPU's share
  # of cfg3 (12 500 windows, three chunks on two engines) and on cfg2, per setting.
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(6_250_000, 30.0, 20260922, 32))
print(bench.product_files(1_000_000, 30.0, 20260921, 32))
PY
  for F in /tmp/racon_amd_cache/files_6250000_30_20260922 /tmp/racon_amd_cache/files_1000000_30_20260921; do
    for q in default 8 16 24; do
      for k in 1 2 3 4; do
        if [ $q = default ]; then env RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/hwq.err" | md5sum | cut -c1-8
        else env GPU_MAX_HW_QUEUES=$q RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/hwq.err" | md5sum | cut -c1-8; fi
        echo "$(basename $F) queues $q: $(grep 'generated consensus' $OUT/hwq.err)"
      done
    done
  done | tee "$OUT/hwq.txt"
fi
if has chunks; then
  # chunk size / engines per device of polish() on one GPU's share of cfg3 (12 500 windows) and on a 25 000-window job
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(6_250_000, 30.0, 20260922, 32))
print(bench.product_files(12_500_000, 30.0, 20260922, 32))
PY
  for F in /tmp/racon_amd_cache/files_6250000_30_20260922 /tmp/racon_amd_cache/files_12500000_30_20260922; do
    for e in "" "RACON_HIP_CHUNK_WINDOWS=3200" "RACON_HIP_CHUNK_WINDOWS=6300" "RACON_HIP_CHUNK_WINDOWS=8400" "RACON_HIP_CHUNK_WINDOWS=12600" "RACON_HIP_ENGINES_PER_DEVICE=3" "RACON_HIP_ENGINES_PER_DEVICE=1"; do
      for k in 1 2 3; do
        env $e RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/chunks.err" | md5sum | cut -c1-8
        echo "$(basename $F) [$e]: $(grep 'generated consensus' $OUT/chunks.err) $(grep -c 'engine .* chunk' $OUT/chunks.err) chunks"
      done
    done
  done | tee "$OUT/chunks.txt"
fi
if has ptrace; then
  # kernel trace of polish() of the binary on one GPU's share of cfg3 (12 500 windows, three chunks on two engines): which
  # consensus kernels overlap, and how long the device has none
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(6_250_000, 30.0, 20260922, 32))
PY
  F=/tmp/racon_amd_cache/files_6250000_30_20260922
  racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2>/dev/null | md5sum
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/ptrace" -o ptrace -- env RACON_HIP_TIMING=1 "$GRAFT_REPO_ROOT/racon_amd/host/racon_hip" -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$GRAFT_REPO_ROOT/$OUT/ptrace.err" | md5sum)
  grep -E "generated consensus|chunk" "$OUT/ptrace.err" | head
  python tools/kernel_timeline.py "$OUT/ptrace" | tee "$OUT/product_kernel_timeline.txt"
  find "$OUT/ptrace" -name "*kernel_trace.csv" -size +2M -delete
fi
if has alignsq; then
  # SQ counters of the pair aligner (k_pair_align) on the align bench: two passes
  n=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"; do
    n=$((n + 1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d "$OUT/asq_$n" -o pmc -- python tools/align_bench.py --reps 1 --host-sample 4 > "$OUT/asq_$n.json" 2> "$OUT/asq_$n.err"
    echo "alignsq pass $n exit $?"
  done
  python - "$OUT" <<'PY' | tee "$OUT/align_sq_summary.txt"
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [set(), 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "asq_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        if "k_pair_align" not in kn: continue
        a = acc[kn][r["Counter_Name"]]; a[0].add(r["Dispatch_Id"]); a[1] += float(r["Counter_Value"] or 0)
for kn, cs in acc.items():
    per = {c: v[1] / max(1, len(v[0])) for c, v in cs.items()}
    print(kn, {c: "%.4g" % v for c, v in sorted(per.items())})
    if "SQ_BUSY_CYCLES" in per and "SQ_INSTS_VALU" in per:
        # SQ_BUSY_CYCLES: quad-cycles summed over the 32 SEs x ...; report ratios that do not need its unit
        ins = per["SQ_INSTS_VALU"] + per["SQ_INSTS_SALU"] + per.get("SQ_INSTS_LDS", 0) + per.get("SQ_INSTS_SMEM", 0) + per.get("SQ_INSTS_BRANCH", 0)
        print("  wave-instructions %.4g (VALU %.3f)" % (ins, per["SQ_INSTS_VALU"] / ins), " per wave %.4g" % (ins / max(1, per.get("SQ_WAVES", 1))))
    if "SQ_WAIT_ANY" in per and "SQ_WAVE_CYCLES" in per:
        print("  wave time: waiting (s_waitcnt) %.3f, issuing %.3f" % (per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"], per.get("SQ_ACTIVE_INST_ANY", 0) / per["SQ_WAVE_CYCLES"]))
PY
fi
if has ngs500; then
  for e in "" "RCN_NO_SMALL=1" "RCN_SMALL_PER_CU=4" ; do
    env $e timeout 900 python bench.py --config ngs_w500 --steps 3 --warmup 1 --no-cpu --no-product --no-upload-leg 2>/dev/null | benchline "ngs_w500 $e"
  done | tee "$OUT/ngs_w500.txt"
fi
if has scratchlimit; then
  # poa_window_kernel2 asks for 416 B of private segment per lane: above the runtime's single-allocation limit scratch is set up per
  # dispatch.  The same lines with the limit raised (HSA_SCRATCH_SINGLE_LIMIT, bytes).
  for e in "" "HSA_SCRATCH_SINGLE_LIMIT=4000000000"; do
    for k in 1 2; do
      env $e timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('cfg2 [$e]', round(j['value']), r['step_kernel_ms'], r.get('launch_ms'))"
    done
    env $e timeout 600 python bench.py --contig 4000000 --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('4mbp [$e]', round(j['value']), r['step_kernel_ms'], r.get('launch_ms'))"
    env $e timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('cfg4 [$e]', round(j['value']), r['step_kernel_ms'], r.get('launch_ms'))"
  done | tee "$OUT/scratch_limit.txt"
fi
