for g in 0 4 8 16 0 8; do
  export LR_GEMM_GROUP_M=$g
  echo "== group_m $g"
  python tools/bench_shapes.py --tiles table --only geglu 2>&1 | grep -v amdgpu | cut -c1-110
  python tools/bench_shapes.py --tiles table --only qkv 2>&1 | grep -v amdgpu | cut -c1-110
  python tools/bench_shapes.py --tiles table --only ff2 2>&1 | grep -v amdgpu | cut -c1-110
done
