#!/bin/bash
# Builds a diagnostic variant of libgmk.so beside the shipped one:  tools/build_variant.sh <name> <extra hipcc flags...>
#   -> generative_models_amd/libgmk_<name>.so   (use with GMK_LIBGMK=generative_models_amd/libgmk_<name>.so)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/generative_models_amd/csrc
obj=/tmp/gmk_variant_$name
mkdir -p $obj
pids=()
for f in gmk_common gn_silu conv_igemm conv_halo conv_wgrad_slots smallconv embed diffusion_ew attention; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off "$@" -c $src/$f.hip -o $obj/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/generative_models_amd/libgmk_$name.so $obj/*.o
echo built generative_models_amd/libgmk_$name.so
