#!/bin/bash
# A/B of stage times (tools/stage_times.py) across library variants, alternating, on one box:
#   tools/ab_stage.sh fp32 "bwdv1 other"      -> product, bwdv1, other, product, bwdv1, other
P=${1:-fp32}
for rep in 1 2; do for L in "" $2; do
  if [ -n "$L" ]; then export VIPNERF_HIP_LIB=$PWD/vip-nerf_amd/lib/libvipnerf_hip_$L.so; else unset VIPNERF_HIP_LIB; fi
  HIP_PRECISION=$P python tools/stage_times.py 2>&1 | tail -1
done; done
