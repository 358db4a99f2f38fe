#!/bin/bash
# Experiment: the deep launch of the split (64 deepest windows, one per CU) on the four-wave DP pipeline over full rows
# (KParams::heavy_ns = 1 for that launch only) instead of the banded one-wave DP.
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
OUT=gpurun_out/${1:-deepwide}; mkdir -p $OUT
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.0f windows/s  step %.2f ms  launches %s  frac %.3f  split %s' % (j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac'], r['split_launch']))"; }
{ run "default" RCN_X=0
  run "deep wide" RCN_SPLIT_DEEP_WIDE=1
  run "deep wide, 2 per CU" RCN_SPLIT_DEEP_WIDE=1 RCN_SPLIT_DEEP_PER_CU=2
  run "deep wide, 2 per CU on 48 CUs" RCN_SPLIT_DEEP_WIDE=1 RCN_SPLIT_DEEP_PER_CU=2 RCN_SPLIT_CUS=48
  run "deep wide, 32 CUs x1" RCN_SPLIT_DEEP_WIDE=1 RCN_SPLIT_CUS=32
  run "default again" RCN_X=0
} > $OUT/deep_wide.txt 2>&1
cat $OUT/deep_wide.txt
RCN_SPLIT_DEEP_WIDE=1 python bench.py --steps 2 --warmup 1 --no-cpu --no-product --no-upload-leg --verify 2>&1 | tail -3
