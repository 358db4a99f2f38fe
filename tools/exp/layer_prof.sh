#!/bin/bash
# per-layer clocks (sink ties, traceback) of the four deepest windows of the bench batch, and the sink-tie totals (profiling build)
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
RCN_PROF_LAYERS=1 RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so python bench.py --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | grep -E "work item|item . layer|sink ties" | cut -c1-300
