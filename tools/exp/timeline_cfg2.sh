#!/bin/bash
# host timeline of the product's polish() on cfg2-shaped files and on one GPU's share of cfg3 (third run of each)
export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
for c, s in ((1_000_000, 20260921), (6_250_000, 20260922)):
    bench.product_files(c, 30.0, s, 32)
PY
for F in /tmp/racon_amd_cache/files_1000000_30_20260921 /tmp/racon_amd_cache/files_6250000_30_20260922; do
  for k in 1 2 3; do
    RCN_DEBUG=1 RACON_HIP_TIMING=1 racon_amd/host/racon_hip -t 32 $F/reads.fastq $F/overlaps.sam $F/targets.fasta 2> /tmp/tl_$k.err | md5sum
  done
  grep -E "racon::Polisher::polish|piece|polish:" /tmp/tl_3.err | head -60
done
