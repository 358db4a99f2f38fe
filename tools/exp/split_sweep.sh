#!/bin/bash
# split launch: how many CUs (= windows, one per CU) the deep launch should get
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.0f windows/s  step %.2f ms  launches %s  frac %.3f' % (j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac']))"; }
run "no split" RCN_SPLIT=0
for c in 16 24 32 40 48 56 64 72; do run "$c CUs x1" RCN_SPLIT_CUS=$c; done
run "32 CUs x2" RCN_SPLIT_CUS=32 RCN_SPLIT_DEEP_PER_CU=2
run "48 CUs x2" RCN_SPLIT_CUS=48 RCN_SPLIT_DEEP_PER_CU=2
run "64 CUs x1 again" RCN_SPLIT_CUS=64
