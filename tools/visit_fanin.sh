#!/bin/bash
# GPU visit: rows with seven / eight in-edges on the move-code path.  band tests, then A/B against the previous build.
set -u
OUT=gpurun_out/${1:-r02c}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_band.py -q -x --timeout 600 > "$OUT/band_tests.log" 2>&1; echo "band tests exit $?" >> "$OUT/band_tests.log"; tail -6 "$OUT/band_tests.log"
bash tools/ab.sh ${1:-r02c} 2 "RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_old.so" "RCN_X=0"
AB_ARGS="--contig 4000000" bash tools/ab.sh ${1:-r02c}_8k 1 "RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_old.so" "RCN_X=0"
bash tools/gpu_round.sh ${1:-r02c} winprof
