#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02i}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_band.py tests/test_gpu_parity.py -q -x --timeout 600 > "$OUT/band_tests.log" 2>&1; echo "band tests exit $?" >> "$OUT/band_tests.log"; tail -25 "$OUT/band_tests.log"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_band.py --deselect tests/test_gpu_parity.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -8 "$OUT/pytest_gpu.log"
bash tools/ab.sh ${1:-r02i} 2 "RCN_NO_BAND=1" "RCN_BAND_SCORES=1" "RCN_X=0"
AB_ARGS="--contig 4000000" bash tools/ab.sh ${1:-r02i}_8k 1 "RCN_NO_BAND=1" "RCN_X=0"
