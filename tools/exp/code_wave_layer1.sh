#!/bin/bash
# DP clocks of the first layer of the four deepest windows (graph = backbone: chain rows only): without and with code waves
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
for V in 1 "" 1 ""; do
echo "== RCN_NO_CODE_WAVE=$V"
env ${V:+RCN_NO_CODE_WAVE=1} RCN_PROF_LAYERS=1 RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so python bench.py --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -E "item [0-3] layer +1:" | cut -c1-120
done
