#!/bin/bash
# Build an experimental libddx.so with extra hipcc flags:  bash profiles/tools/build_variant.sh <name> <flags...>
# -> profiles/tools/variants/libddx_<name>.so   (use with DDX_LIB=...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/doubletdetection_amd/csrc
out=$root/profiles/tools/variants
obj=$out/obj_$name
mkdir -p "$obj"
for f in ddx_api k_sparse k_pca k_bitplane k_knn k_prologue k_louvain; do
    extra=""; [ $f = k_knn ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -c $src/$f.hip -o $obj/$f.o &
done
for f in louvain hostmath; do g++ -O2 -std=c++17 -fPIC -ffp-contract=off -pthread -c $src/$f.cpp -o $obj/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $out/libddx_$name.so $obj/*.o
echo $out/libddx_$name.so
