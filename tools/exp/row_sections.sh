#!/bin/bash
# (needs tools/exp/patches/row_sections.patch applied and racon_amd/csrc built with -DRCN_PROF_SECT as libracon_hip_sect.so)
# clocks per section of a chain / one-predecessor row of the banded DP (RCN_PROF_SECT build): the bench batch as launched
# (deep windows with code waves: HELP variant; all others: the plain variant), then every window alone on a CU without code waves
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
L=$PWD/racon_amd/csrc/libracon_hip_sect.so
echo "== bench batch as launched"
RACON_HIP_LIB=$L python bench.py --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep "rows of the banded DP"
echo "== 200 windows, one per CU, no code waves"
RCN_NO_CODE_WAVE=1 RCN_SPLIT=0 RCN_WG_PER_CU=1 RACON_HIP_LIB=$L python bench.py --contig 100000 --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep "rows of the banded DP"
echo "== 200 windows, one per CU, code waves"
RCN_SPLIT=0 RCN_WG_PER_CU=1 RACON_HIP_LIB=$L python bench.py --contig 100000 --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep "rows of the banded DP"
echo "== 200 windows, eight per CU"
RCN_SPLIT=0 RCN_WG_PER_CU=8 RACON_HIP_LIB=$L python bench.py --contig 100000 --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep "rows of the banded DP"
