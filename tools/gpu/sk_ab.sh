for mode in 0 1 0 1; do
  export LEFTREFILL_SPLITK_MODE=$mode
  echo "== splitk_mode $mode"
  python tools/bench_gemm.py 1024 1280 11520 9 256 160 8 0 2>&1 | grep tile
  python tools/bench_gemm.py 1024 1280 23040 9 256 160 8 8 2>&1 | grep tile
  python tools/bench_gemm.py 4096 1280 11520 9 256 320 4 8 2>&1 | grep tile
  python tools/bench_gemm.py 4096 1280 23040 9 256 320 4 8 2>&1 | grep tile
  python tools/bench_gemm.py 16384 640 17280 9 256 320 2 8 2>&1 | grep tile
done
