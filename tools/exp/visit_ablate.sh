#!/bin/bash
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
set -u
OUT=gpurun_out/${1:-r02h}; mkdir -p "$OUT"
export TMPDIR=/tmp
VARS=()
for k in 0 8 11 10; do VARS+=("RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_abl$k.so"); done
bash tools/ab.sh ${1:-r02h} 2 "${VARS[@]}" | sed -e "s#RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_##"
# HBM traffic of the banded and the unbanded kernel
BENCH_PROF="python bench.py --steps 3 --warmup 1 --no-cpu"
for V in noband band; do
  if [ $V = noband ]; then export RCN_NO_BAND=1; else unset RCN_NO_BAND; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/${V}/pmc_$C" -o pmc -- $BENCH_PROF > "$OUT/${V}_pmc_$C.json" 2> "$OUT/${V}_pmc_$C.err"
  done
  python tools/pmc_summary.py "$OUT/$V" | grep -E "poa_window|traffic"
done
unset RCN_NO_BAND
