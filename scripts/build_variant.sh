#!/bin/bash
# Kernel experiments: build pagnerf_amd/lib/libpagnerf_hip_<tag>.so from re-compiled source(s) (+ the objects of the regular build)
#   bash scripts/build_variant.sh <tag> <source[,source...]> "<extra flags>"       then run with PAG_LIB_VARIANT=<tag>
# (cross-compiles here without a GPU; the .so travels to the GPU box with the snapshot)
set -e
tag=$1; srcs=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/pagnerf_amd/lib
declare -A rebuilt
for src in ${srcs//,/ }; do
  base=$(basename $src .hip)
  extra=""
  case $base in encode|render|assign|loss|optim|pose|regularizer) extra="-ffp-contract=off";; mlp) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-value $extra $flags -c $root/pagnerf_amd/csrc/$base.hip -o $lib/obj/${base}_$tag.o &
  rebuilt[$base]=1
done
wait
objs=""
for o in api encode render mlp assign loss optim pose regularizer; do
  if [ -n "${rebuilt[$o]}" ]; then objs="$objs $lib/obj/${o}_$tag.o"; else objs="$objs $lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/libpagnerf_hip_$tag.so $objs
rm -f $lib/obj/*_$tag.o      # the snapshot that travels to the GPU box carries lib/: keep it small
echo built $lib/libpagnerf_hip_$tag.so
