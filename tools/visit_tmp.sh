#!/bin/bash
set -u
TAG=${1:-r02t}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
L=$PWD/racon_amd/csrc
timeout 600 python -m pytest tests/test_gpu_band.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -x --timeout 600 > "$OUT/tests.log" 2>&1; echo "tests exit $?" >> "$OUT/tests.log"; tail -3 "$OUT/tests.log"
bash tools/ab.sh $TAG 3 "RACON_HIP_LIB=$L/libracon_hip_old.so" "RCN_X=0" | sed -e "s#RACON_HIP_LIB=$L/libracon_hip_##"
