#!/bin/bash
# Build a variant of the library for a same-box A/B:  tools/build_variant.sh NAME [extra hipcc flags, e.g. -DXA_SPREAD=0]
# -> leftrefill_amd/lib/variants/libleftrefill_hip_NAME.so ; run with LEFTREFILL_LIB_PATH=<that file>.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=leftrefill_amd/lib/variants; mkdir -p $out /tmp/lrv_$name
objs=""
for s in norm elementwise gemm_conv attention attention_bwd xattn_block ffn_block conv_out; do
  extra=""
  case $s in attention|attention_bwd|xattn_block|ffn_block) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra "$@" -c leftrefill_amd/csrc/$s.hip -o /tmp/lrv_$name/$s.o &
  objs="$objs /tmp/lrv_$name/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libleftrefill_hip_$name.so $objs
echo $out/libleftrefill_hip_$name.so
