#!/bin/bash
# granularity of the band window's offsets (kBandG: 32 in the product): fewer, larger window shifts against a worse-centred window
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-product --no-upload-leg"
for L in libracon_hip libracon_hip_g64 libracon_hip_g128 libracon_hip libracon_hip_g64 libracon_hip_g128; do
echo "== $L"; RACON_HIP_LIB=$PWD/racon_amd/csrc/$L.so $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.0f windows/s  step %.2f ms  launches %s  banded %s redone %s %s' % (j['value'], r['step_kernel_ms'], ['%.2f' % v for v in r['launch_ms']], r['banded_alignments'], r['band_redone'], r['band_redo_why']))"
done
