# graph-timed check of the pipelined GEMM instances on representative shapes (one line per shape / plan)
for c in "512 1280 2560 1 128 160 1 4" "4096 1280 1280 1 128 160 1 4" "2048 1280 1280 1 128 128 1 4" "2048 1280 1280 1 128 160 1 4" "1024 1280 11520 9 128 160 2 4" "4096 1280 11520 9 128 160 1 4" "16384 640 640 1 256 160 1 0" "16384 640 5760 9 256 160 1 0" "65536 320 2880 9 256 320 1 0" "16384 1920 640 1 256 160 1 0" "65536 320 320 1 256 160 1 0" "4096 3840 1280 1 128 160 1 4"; do
  python tools/bench_gemm.py $c --reps 40 --graph 2>&1 | tail -1
done
python tools/bench_gemm.py 4096 10240 1280 1 256 320 1 0 --geglu --reps 40 --graph 2>&1 | tail -1
