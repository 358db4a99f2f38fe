#!/bin/bash
# cfg2 bench lines: shipped library, variants (libracon_hip_v_<name>.so ...), and the shipped one without code waves
export RCN_EXPERIMENT=1
QB="--steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-40s %.0f windows/s  step %.2f ms  launches %s  frac %.3f' % ('$1', j.get('value_kernel_leg', j['value']), r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['frac']))"; }
for k in 1 2; do
  python bench.py $QB 2>/dev/null | line "shipped ($k)"
  for v in "$@"; do RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_v_$v.so python bench.py $QB 2>/dev/null | line "$v ($k)"; done
  RCN_NO_CODE_WAVE=1 python bench.py $QB 2>/dev/null | line "shipped, RCN_NO_CODE_WAVE=1 ($k)"
done
