# tile / split sweep of the tile-starved short-K shapes, timed as hipGraph replays (tools/bench_gemm.py --graph)
for shape in "2048 1280 1280 1" "4096 1280 1280 1" "512 1280 1280 1" "1024 1280 1280 1" "8192 640 640 1" "16384 640 640 1" "32768 320 320 1" "2048 3840 1280 1" "2048 1280 5120 1"; do
  for cfg in "128 64 1 0" "128 128 1 0" "128 160 1 0" "128 128 1 4" "128 160 1 4" "128 160 2 4" "128 64 2 0" "256 160 1 0" "256 128 1 0"; do
    python tools/bench_gemm.py $shape $cfg --reps 40 --graph 2>&1 | tail -1
  done
done
