# GPU-side fixed cost and per-K-step slope of the small-M GEMM instances (hipGraph replays)
for M in 512 4096; do for K in 64 320 640 1280 2560; do
  for cfg in "128 64 1 0" "128 128 1 4" "128 160 1 4"; do
    python tools/bench_gemm.py $M 1280 $K 1 $cfg --reps 40 --graph 2>&1 | tail -1
  done
done; done
