#!/bin/bash
# Builds libvipnerf_hip.so for gfx950 in-tree (vip-nerf_amd/lib/).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/lib"
mkdir -p "$OUT" "$OUT/obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result ${VIPNERF_EXTRA_FLAGS}"
pids=()
for f in vipnerf_pack_bf16n vipnerf_mlp_fwd_bf16n vipnerf_mlp_bwd_bf16n vipnerf_mlp_bwd_f32 vipnerf_mlp_fwd_f32 vipnerf_mlp_fwd_pt2 vipnerf_mlp_bwd_pt2 vipnerf_wgrad vipnerf_wgrad16 vipnerf_ray vipnerf_camera vipnerf_psv vipnerf_debug vipnerf_generic vipnerf_api; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/obj/$f.o" ] || [ -n "$(find "$SRC" "$HERE/../include" -name '*.h' -newer "$OUT/obj/$f.o")" ]; then
    ( hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/obj/$f.o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC "$OUT"/obj/*.o -o "$OUT/libvipnerf_hip.so"
echo "built $OUT/libvipnerf_hip.so"
