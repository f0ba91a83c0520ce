#!/bin/bash
# A library variant next to the product one: tools/build_variant.sh NAME "FLAGS" [file ...]
#   -> vip-nerf_amd/lib/libvipnerf_hip_NAME.so   (select with VIPNERF_HIP_LIB=...)
# Only the listed translation units (default: the two-point-tile MLP kernels, the 16-bit weight gradients and the API) are recompiled with
# FLAGS; the rest are the product build's objects (run vip-nerf_amd/build.sh first).  vipnerf_api is always recompiled, so
# vipnerf_build_info() of the variant reports FLAGS.
set -e
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-"vipnerf_mlp_fwd_pt2 vipnerf_mlp_bwd_pt2 vipnerf_wgrad16"}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$ROOT/vip-nerf_amd/csrc; LIB=$ROOT/vip-nerf_amd/lib; OUT=$LIB/var_$NAME
mkdir -p $OUT
cp $LIB/obj/*.o $OUT/
for f in $FILES vipnerf_api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $FLAGS -c $SRC/$f.hip -o $OUT/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $LIB/libvipnerf_hip_$NAME.so
rm -rf $OUT
echo "built libvipnerf_hip_$NAME.so ($FLAGS)"
