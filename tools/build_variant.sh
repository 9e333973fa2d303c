#!/bin/bash
# CPU (this container): builds makeupdiffuse_amd/libmkd_<name>.so from the in-tree sources with extra compiler flags, for A/B
# experiments on the GPU box (select it there with MKD_LIB_PATH).    tools/build_variant.sh ablate -DMKD_EXP_ABLATE
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/makeupdiffuse_amd/csrc
obj=$src/_obj/variant_$name
mkdir -p $obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=fast"
pids=()
for f in kernels_gemm kernels_conv kernels_norm kernels_attn kernels_tfm kernels_misc engine; do
  hipcc $FLAGS "$@" -c $src/$f.hip -o $obj/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/makeupdiffuse_amd/libmkd_$name.so $obj/*.o
echo $root/makeupdiffuse_amd/libmkd_$name.so
