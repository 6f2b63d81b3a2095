# hot (one weight buffer) vs cold (120 weight copies visited in turn, > 256 MB of Infinity Cache) launches, graph-timed, reps = 120
for c in "4096 1280 1280 1 128 160 1 4" "1024 1280 1280 1 128 64 1 0" "16384 640 640 1 256 160 1 0" "4096 1280 11520 9 128 160 1 4" "1024 1280 11520 9 128 160 2 4" "16384 640 5760 9 256 160 1 0"; do
  echo -n "hot   "; python tools/bench_gemm.py $c --reps 120 --graph 2>&1 | tail -1
  echo -n "coldW "; python tools/bench_gemm.py $c --reps 120 --graph --cold 120 2>&1 | tail -1
  echo -n "coldWX "; python tools/bench_gemm.py $c --reps 120 --graph --cold 120 --cold-x 2>&1 | tail -1
done
