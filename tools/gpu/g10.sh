cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/chain_proto tools/micro/chain_stream_proto.hip 2>/dev/null && /tmp/chain_proto > gpurun_out/r4/g10_chain_proto.txt 2>&1
timeout 900 python tools/bench_shapes.py --tiles table > gpurun_out/r4/g10_shapes_base.txt 2>&1
LEFTREFILL_LIB_PATH=$PWD/leftrefill_amd/lib/variants/libleftrefill_hip_setprio.so timeout 900 python tools/bench_shapes.py --tiles table > gpurun_out/r4/g10_shapes_setprio.txt 2>&1
timeout 900 python tools/bench_shapes.py --tiles table >> gpurun_out/r4/g10_shapes_base.txt 2>&1
timeout 1200 bash tools/pmc_attn_r4.sh $PWD/gpurun_out/r4/pmc_attn > gpurun_out/r4/g10_pmc_attn.log 2>&1
echo done
