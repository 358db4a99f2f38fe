#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02g}; mkdir -p "$OUT"
L=$PWD/racon_amd/csrc
for m in 2 3; do
  RACON_HIP_LIB=$L/libracon_hip_sm$m.so timeout 600 python -m pytest tests/test_gpu_band.py tests/test_gpu_parity.py -q -x --timeout 600 > "$OUT/tests_sm$m.log" 2>&1; echo "store mode $m tests exit $?"; tail -2 "$OUT/tests_sm$m.log"
done
bash tools/ab.sh ${1:-r02g} 2 "RCN_NO_BAND=1" "RCN_X=0" "RACON_HIP_LIB=$L/libracon_hip_sm1.so" "RACON_HIP_LIB=$L/libracon_hip_sm2.so" "RACON_HIP_LIB=$L/libracon_hip_sm3.so" | sed -e "s#RACON_HIP_LIB=$L/libracon_hip_##"
