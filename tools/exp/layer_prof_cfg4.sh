#!/bin/bash
# per-layer clocks (DP, sink ties, traceback) of the first work items of the cfg4 batch (profiling build)
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
RCN_PROF_LAYERS=1 RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so python bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | grep -E "work item|item . layer|sink ties|Subgraph" | cut -c1-200
