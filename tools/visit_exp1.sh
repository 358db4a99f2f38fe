#!/bin/bash
# GPU visit (experiments): what the code store costs the DP; what a deep window gains from fewer co-resident windows.
set -u
TAG=${1:-r02d}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
L=$PWD/racon_amd/csrc
bash tools/ab.sh $TAG 2 "RCN_X=0" "RACON_HIP_LIB=$L/libracon_hip_cs1.so" "RACON_HIP_LIB=$L/libracon_hip_cs2.so" "RACON_HIP_LIB=$L/libracon_hip_cs3.so" | sed -e "s#RACON_HIP_LIB=$L/libracon_hip_##"
for S in 2048 1536 1024; do
  echo "== slots $S"
  RACON_HIP_LIB=$L/libracon_hip_prof.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --slots $S > "$OUT/slots_$S.json" 2> "$OUT/slots_$S.txt"
  python -c "
import json,sys
j=json.loads(open('$OUT/slots_$S.json').read().strip().splitlines()[-1]); print('launch %.2f ms  %.0f w/s' % (j['roofline']['avg_launch_ms'], j['value']))"
  grep -A4 "per-window clocks" "$OUT/slots_$S.txt" | tail -5
done
