#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02f}; mkdir -p "$OUT"
VARS=()
for k in 0 2 8 9; do VARS+=("RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_abl$k.so"); done
bash tools/ab.sh ${1:-r02f} 2 "${VARS[@]}" | sed -e "s#RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_##"
