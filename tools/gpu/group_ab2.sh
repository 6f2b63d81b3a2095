for g in auto 0 auto 0; do
  if [ $g = auto ]; then unset LR_GEMM_GROUP_M; else export LR_GEMM_GROUP_M=$g; fi
  echo "== group_m $g"
  python tools/bench_shapes.py --tiles table --only "conv3x3" 2>&1 | grep best | cut -c1-100
  python tools/bench_gemm.py 4096 1280 11520 9 256 320 4 0 | grep tile
  python tools/bench_gemm.py 4096 1280 14080 9 256 320 4 0 | grep tile
  python tools/bench_gemm.py 65536 640 5760 9 256 320 1 0 | grep tile
  python tools/bench_gemm.py 16384 1280 11520 9 256 160 1 0 | grep tile
  python tools/bench_gemm.py 16384 640 640 1 256 160 1 0 | grep tile
  python tools/bench_gemm.py 65536 960 320 1 256 320 1 0 | grep tile
done
