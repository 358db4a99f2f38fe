#!/bin/bash
# Round-6 GPU visits.  Usage: bash tools/gpu_round6.sh <tag> [what...]
#   what: probe (tools/probe/row_chain.hip) parity (the parity tests that exercise the banded DP) tests (whole GPU suite)
#         ab (cfg2 / 4 Mbp / w1000 bench lines, shipped library against libracon_hip_v_*.so variants) bench (default line)
#         fuzz (tools/fuzz_sweep.py) final (kernel stats + counter passes + all lines)
set -u
TAG=${1:-r06a}; shift || true
WHAT=${*:-probe parity ab}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export RCN_EXPERIMENT=1
has() { [[ " $WHAT " == *" $1 "* ]]; }
# the libraries travel with the tree as they were built: refuse to measure a library that is older than its sources (a closing set of this round
# once ran a small-window kernel that had been reverted in the sources but not rebuilt: the stamped source hash then says nothing)
for d in racon_amd/csrc racon_amd/host; do make -q -C $d all 2>/dev/null || { echo "$d: the built library is older than its sources -- make -C $d first" | tee "$OUT/STALE_LIBRARY.txt"; exit 1; }; done
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showproductname 2>/dev/null | head -12) > "$OUT/box.txt" 2>&1
benchline() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%s: %.0f windows/s  step %.2f ms  launches %s  frac %.3f  gcups %.0f  small %s bailed %s' % ('$1', j.get('value_kernel_leg', j['value']), r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac'], r.get('gcups',0), r.get('small_windows'), r.get('small_bailed')))"; }
QB="--steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"

if has probe; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/row_chain tools/probe/row_chain.hip && /tmp/row_chain | tee "$OUT/row_chain_probe.txt"
fi
if has parity; then
  timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_band.py tests/test_gpu_fuzz.py "tests/test_gpu_fullsize.py" \
      -m gpu -q -x --durations=10 > "$OUT/pytest_parity.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_parity.log"; tail -25 "$OUT/pytest_parity.log"
fi
if has tests; then
  timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -30 "$OUT/pytest_gpu.log"
fi
if has new; then
  timeout 3000 python -m pytest tests/test_gpu_selfcheck.py tests/test_cli_e2e.py tests/test_gpu_product_path.py tests/test_gpu_pair_align.py tests/test_gpu_bench_contract.py \
      -m gpu -q -x --durations=12 > "$OUT/pytest_new.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_new.log"; tail -30 "$OUT/pytest_new.log"
fi
if has align; then
  # the device pairwise aligner: its parity tests, the cfg2-shaped bench line (3000 overlaps), TCUPS at size come with `shardtime`
  timeout 1500 python -m pytest tests/test_gpu_pair_align.py -m gpu -q -x --durations=5 > "$OUT/pytest_align.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_align.log"; tail -12 "$OUT/pytest_align.log"
  timeout 900 python tools/align_bench.py > "$OUT/align_bench.json" 2> "$OUT/align_bench.err"; echo "align_bench exit $?"; cut -c1-900 "$OUT/align_bench.json"
fi
if has shardtime; then
  # where a shard's time goes when a job is cut into more window ranges than devices (cfg5 at a quarter of its size, four sequential shards)
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from racon_amd.synth import simulate_fragment_files
SC = float(os.environ.get("SHARD_SCALE", "0.25")); d = "/tmp/racon_amd_cache/cfg5_%g" % SC
if not os.path.exists(d + "/.done"):
    p = simulate_fragment_files(d, int(33_333_333 * SC), int(100_000 * SC), seed=20260924); open(d + "/.done", "w").write(str(p["n_overlaps"]))
PY
  F=/tmp/racon_amd_cache/cfg5_${SHARD_SCALE:-0.25}
  for PIPE in ${SHARD_PIPES:-none}; do
    [ "$PIPE" = none ] && PIPE=""
    env $PIPE RACON_HIP_DEVICE_SHARDS=${SHARD_N:-4} RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -f -t 32 --cudaaligner-batches 1 $F/reads.fastq $F/overlaps.paf $F/reads.fastq 2> "$OUT/shardtime.err" | md5sum
    echo "== ${PIPE:-default}"; grep -E "racon::|racon_hip\] (self|pairs|pair aligner|run:)" "$OUT/shardtime.err" | grep -v "piece\|collect\|pass of" | cut -c1-330 | tee "$OUT/shardtime_${PIPE:+device_pipeline}.txt" | tail -30
  done
fi
if has cfg5; then
  timeout ${CFG5_TIMEOUT:-2400} python tools/cfg5_whole.py ${CFG5_ARGS:---scale 1.0 --shards 8} --record-md5 "$OUT/cfg5_record_md5.npy" > "$OUT/cfg5_whole.json" 2> "$OUT/cfg5_whole.err"
  echo "cfg5 exit $?"; tail -3 "$OUT/cfg5_whole.err" | cut -c1-400; cut -c1-4500 "$OUT/cfg5_whole.json"
fi
if has cfg3bin; then
  timeout 2400 python bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu --no-upload-leg --product-contig 0 > "$OUT/bench_cfg3_1gpu.json" 2> "$OUT/bench_cfg3.err"
  echo "cfg3 exit $?"; cut -c1-2500 "$OUT/bench_cfg3_1gpu.json"; tail -3 "$OUT/bench_cfg3.err"
fi
if has inittime; then
  # initialize() by phase at size (the Logger's own lines + RACON_HIP_TIMING): cfg3 whole (50 Mbp: 1.5 GB FASTQ + 3 GB SAM) and cfg5 whole
  # (-f, 2 GB FASTQ that is reads and targets, 4.8 M PAF overlaps left to the device aligner); second run of each = warm page cache
  python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.product_files(50_000_000, 30.0, 20260922, 32))
PY
  F=/tmp/racon_amd_cache/files_50000000_30_20260922
  for k in 1 2; do
    RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> "$OUT/inittime_cfg3_$k.err" | md5sum | cut -c1-8
    grep -E "racon::Polisher::(initialize|polish|)\]" "$OUT/inittime_cfg3_$k.err" | grep -v "chunk\|shard " | cut -c1-200
  done 2>&1 | tee "$OUT/inittime_cfg3.txt"
  G=/tmp/racon_amd_cache/cfg5_1
  if [ -f $G/reads.fastq ]; then
    for k in 1 2; do
      RACON_HIP_DEVICE_SHARDS=8 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -f -t 32 --cudaaligner-batches 1 $G/reads.fastq $G/overlaps.paf $G/reads.fastq 2> "$OUT/inittime_cfg5_$k.err" | md5sum | cut -c1-8
      grep -E "racon::Polisher::(initialize|polish|)\]" "$OUT/inittime_cfg5_$k.err" | cut -c1-330
    done 2>&1 | tee "$OUT/inittime_cfg5.txt"
  fi
fi
if has twice; then
  # two Polishers one after the other in ONE process (bench.py's product leg does that): what the second one pays, by phase
  RACON_HIP_TIMING=1 RCN_DEBUG=1 python - > "$OUT/twice.txt" 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from racon_amd.polisher import Polisher
f = bench.product_files(1_000_000, 30.0, 20260921, 32)
os.environ["RACON_HIP_DEVICE_WINDOWS"] = "auto"
for k in range(3):
    sys.stderr.write("======== Polisher %d\n" % k); sys.stderr.flush()
    p = Polisher(f["reads"], f["sam"], f["targets"], "kC", 500, 10.0, 0.3, True, 3, -5, -4, 32, 1)
    t = time.perf_counter(); p.initialize(); ti = time.perf_counter() - t
    p.polish(True)
    sys.stderr.write("======== initialize %.3f s, polish %.4f s\n" % (ti, p.polish_seconds())); sys.stderr.flush()
    t = time.perf_counter(); p.close(); sys.stderr.write("======== close %.3f s\n" % (time.perf_counter() - t)); sys.stderr.flush()
PY
  grep -E "========|racon::Polisher::initialize\]|timing" "$OUT/twice.txt" | cut -c1-260
fi
if has ab; then
  for k in 1 2; do
    python bench.py $QB 2>/dev/null | benchline "cfg2 as shipped ($k)"
    for v in ${VARIANTS:-nooct}; do [ -f racon_amd/csrc/libracon_hip_v_$v.so ] && RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_v_$v.so python bench.py $QB 2>/dev/null | benchline "cfg2 libracon_hip_v_$v.so ($k)"; done
  done 2>&1 | tee "$OUT/ab.txt"
  python bench.py --contig 4000000 $QB 2>/dev/null | benchline "4 Mbp as shipped" | tee -a "$OUT/ab.txt"
  for v in ${VARIANTS:-nooct}; do [ -f racon_amd/csrc/libracon_hip_v_$v.so ] && RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_v_$v.so python bench.py --contig 4000000 $QB 2>/dev/null | benchline "4 Mbp libracon_hip_v_$v.so" | tee -a "$OUT/ab.txt"; done
  python bench.py --config w1000 $QB 2>/dev/null | benchline "w1000 as shipped" | tee -a "$OUT/ab.txt"
  python bench.py --config cfg4 $QB 2>/dev/null | benchline "cfg4 as shipped" | tee -a "$OUT/ab.txt"
fi
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
if has bench; then
  timeout 1500 python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench exit $?"; cut -c1-3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if has fuzz; then
  timeout 1500 python tools/fuzz_sweep.py --seeds ${FUZZ_SEEDS:-120} --out "$OUT/fuzz_sweep.json" > "$OUT/fuzz_sweep.log" 2>&1
  echo "fuzz exit $?"; tail -4 "$OUT/fuzz_sweep.log" | cut -c1-1500
  timeout 1500 python tools/fuzz_sweep.py --no-small --first-seed 9000 --seeds ${FUZZ_K2_SEEDS:-60} --out "$OUT/fuzz_sweep_kernel2.json" > "$OUT/fuzz_sweep_kernel2.log" 2>&1
  echo "fuzz (kernel2) exit $?"; tail -2 "$OUT/fuzz_sweep_kernel2.log" | cut -c1-1200
fi
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT" | tail -1
