#!/bin/bash
# Build a DEVELOPER variant of the library for a same-box A/B:  [VARIANT_SRCS="conv_halo gemm_conv"] tools/build_variant.sh NAME [extra hipcc flags]
# -> leftrefill_amd/lib/variants/libleftrefill_hip_NAME.so ; run with LEFTREFILL_LIB_PATH=<that file>.
# Every variant is compiled with -DLR_DEV_VARIANTS: the measured-and-lost kernels (ping-pong attention, in-launch split-K reduce, alternative
# tile orders, GroupNorm fold) and the LR_* knob table (lr_dev_set; leftrefill_amd/_lib.py forwards LR_* environment variables) exist in
# these builds only -- the product library (python -m leftrefill_amd.build) has no switches.  VARIANT_SRCS must then include elementwise
# (the knob table) and every source whose knobs are used; the default recompiles everything.
# VARIANT_SRCS: recompile only these sources with the extra flags and link the product build's objects for the rest (run
# `python -m leftrefill_amd.build` first); default: every source.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=leftrefill_amd/lib/variants; mkdir -p $out /tmp/lrv_$name
all=$(python -c "from leftrefill_amd import build; print(' '.join(s[:-4] for s in build.SOURCES))")
objs=""
for s in $all; do
  if [ -n "$VARIANT_SRCS" ] && ! echo " $VARIANT_SRCS " | grep -q " $s "; then objs="$objs leftrefill_amd/build/$s.o"; continue; fi
  extra=""
  case $s in attention|attention_bwd|xattn_block|ffn_block|stin_block|rowlin) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DLR_DEV_VARIANTS $extra "$@" -c leftrefill_amd/csrc/$s.hip -o /tmp/lrv_$name/$s.o &
  objs="$objs /tmp/lrv_$name/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libleftrefill_hip_$name.so $objs
echo $out/libleftrefill_hip_$name.so
