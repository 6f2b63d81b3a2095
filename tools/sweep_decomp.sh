# timing decomposition of the 128-row deep-ring GEMM instance on a lone-block shape and a full-chip shape (graph-timed): product build vs
# builds without MFMAs / without LDS-DMA / without fragment reads (tools/build_variant.sh nomfma -DLR_GEMM_NO_MFMA etc.; results are garbage, only time counts)
for lib in "" nomfma nodma noreads nodma_noreads; do
  if [ -n "$lib" ]; then export LEFTREFILL_LIB_PATH=$PWD/leftrefill_amd/lib/variants/libleftrefill_hip_$lib.so; else unset LEFTREFILL_LIB_PATH; fi
  for shape in "512 1280 2560 1" "4096 1280 1280 1" "2048 1280 1280 1"; do
    for cfg in "128 160 1 4" "128 128 1 4"; do
      echo -n "[${lib:-product}] "; python tools/bench_gemm.py $shape $cfg --reps 40 --graph 2>&1 | tail -1
    done
  done
done
