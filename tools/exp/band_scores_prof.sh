#!/bin/bash
# DP clocks of the deepest windows with move codes (default) and with the banded DP storing scores (RCN_BAND_SCORES=1: no
# argmax tracking, no code assembly in the row): what the DP wave would cost if another wave assembled the codes
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
for V in "" 1; do
echo "== RCN_BAND_SCORES=$V"
env ${V:+RCN_BAND_SCORES=1} RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so python bench.py --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | grep -E "work item" | cut -c1-300
done
