#!/bin/bash
# the code waves (poa_band.hpp, HELP): A/B on the bench batch, parity of all 2000 windows, and the whole-launch variant
# (RCN_WG_PER_CU=1: every window alone on a CU -> every banded alignment with code waves) on the test suite's parity files
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
run() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.0f windows/s  step %.2f ms  launches %s  frac %.3f  redone %s %s' % (j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac'], r['band_redone'], r['band_redo_why']))"; }
run "code waves off" RCN_NO_CODE_WAVE=1
run "code waves (default)" RCN_X=0
run "code waves, 64 CUs" RCN_SPLIT_CUS=64
run "code waves off again" RCN_NO_CODE_WAVE=1
run "code waves again" RCN_X=0
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-product --no-upload-leg --verify 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('verified', j.get('verified_windows'), j.get('verified_flags'), j['value'])"
RCN_SPLIT=0 RCN_WG_PER_CU=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_band.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -30
