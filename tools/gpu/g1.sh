set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
( ls /sys/class/drm/ ; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d; for f in freq1_input freq1_label freq2_input power1_average power1_input power1_cap temp1_input; do echo -n "$f: "; cat $d/$f 2>&1; done; done; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -20; which rocm-smi amd-smi; timeout 60 rocm-smi --showclocks --showpower --json 2>&1 | head -c 3000; echo; timeout 60 amd-smi metric --clock --power --json 2>&1 | head -c 3000 ) > gpurun_out/r4/sysfs_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak > gpurun_out/r4/mfma_peak.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -15 > gpurun_out/r4/test_attn.txt
timeout 600 python tools/bench_attn_pp.py 5 > gpurun_out/r4/bench_attn_pp.txt 2>&1
timeout 600 python tools/trace_gemm.py --tiles=256x160,128x160x4,256x320 "l1 KC" "l2 KC" "l0 KC resid+stats" "l1 ff2" > gpurun_out/r4/trace_kc.txt 2>&1
tail -5 gpurun_out/r4/*.txt
