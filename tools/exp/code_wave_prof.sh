#!/bin/bash
# phase clocks of the deepest windows with and without the code waves (profiling build)
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
for V in 1 ""; do
echo "== RCN_NO_CODE_WAVE=$V"
env ${V:+RCN_NO_CODE_WAVE=1} RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so python bench.py --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | grep -E "work item|code waves" | cut -c1-300
done
