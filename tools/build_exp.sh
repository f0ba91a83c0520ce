#!/bin/bash
# build experiment variants: tools/build_exp.sh N  -> vip-nerf_amd/lib/libvipnerf_hip_expN.so
set -e
N=$1
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$ROOT/vip-nerf_amd/csrc
OUT=$ROOT/vip-nerf_amd/lib/exp$N
mkdir -p $OUT
for f in vipnerf_pack vipnerf_pack_bf16 vipnerf_pack_bf16n vipnerf_mlp_fwd vipnerf_mlp_fwd_bf16 vipnerf_mlp_bwd_bf16 vipnerf_mlp_fwd_bf16n vipnerf_mlp_bwd_bf16n vipnerf_mlp_fwd_pt2 vipnerf_mlp_bwd_pt2 vipnerf_mlp_bwd vipnerf_wgrad vipnerf_wgrad16 vipnerf_ray vipnerf_camera vipnerf_psv vipnerf_debug vipnerf_generic vipnerf_api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DVN_EXP=$N $VN_EXTRA -c $SRC/$f.hip -o $OUT/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $ROOT/vip-nerf_amd/lib/libvipnerf_hip_exp$N.so
rm -rf $OUT
echo built exp$N
