export RCN_EXPERIMENT=1
for L in prof prof_nooct; do
echo "== libracon_hip_$L.so"
RACON_HIP_LIB=$PWD/racon_amd/csrc/libracon_hip_$L.so python bench.py --steps 1 --warmup 0 --no-cpu --no-product --no-upload-leg 2>&1 >/dev/null | grep -v amdgpu.ids | grep -E "work item|per-window|code waves|sink ties" | cut -c1-300
done
