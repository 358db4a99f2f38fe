export RCN_EXPERIMENT=1   # engine.hip read_knobs: RCN_* switches are ignored without it
for W in "" 1; do
echo "== RCN_SPLIT_DEEP_WIDE=$W"
RCN_SPLIT_DEEP_WIDE=$W RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_prof.so python bench.py --steps 1 --warmup 1 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | tail -6 | cut -c1-300
done
