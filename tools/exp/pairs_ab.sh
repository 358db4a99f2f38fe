#!/bin/bash
# row pairs of the banded DP (meta bit 31): A/B against RCN_NO_ROW_PAIRS=1, all 2000 windows against the oracle
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.0f windows/s  step %.2f ms  launches %s  frac %.3f  redone %s code waves %s' % (j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac'], r['band_redone'], r.get('code_wave_alignments')))"; }
run "no pairs" RCN_NO_ROW_PAIRS=1
run "pairs" RCN_X=0
run "pairs" RCN_X=0
run "no pairs" RCN_NO_ROW_PAIRS=1
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-product --no-upload-leg --verify 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('verified', j.get('verified_windows'), j.get('verified_flags'), j['value'])"
