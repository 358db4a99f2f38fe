#!/bin/bash
# in-situ clocks per row class of the banded DP (RCN_PROF_ROWS build), whole launch at one work-group per CU, with and without the code waves
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
for V in 1 ""; do
echo "== RCN_NO_CODE_WAVE=$V"
env ${V:+RCN_NO_CODE_WAVE=1} RCN_SPLIT=0 RCN_WG_PER_CU=1 RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_rows.so python bench.py --contig 100000 --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | grep -E "rows|clocks" | cut -c1-200
done
