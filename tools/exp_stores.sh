#!/bin/bash
# Store-ablation A/B of the training kernels (timing-only builds 40 / 41 / 42, vipnerf_bf16n.h): tools/exp_stores.sh "fp16 bf16"
# prints per-stage device times for each build, then WRITE_SIZE of the default and the no-encoding-stores build.
ROOT=$(pwd)
for P in $1; do
  for L in "" 40 41 42; do
    if [ -n "$L" ]; then export VIPNERF_HIP_LIB=$ROOT/vip-nerf_amd/lib/libvipnerf_hip_exp$L.so; else unset VIPNERF_HIP_LIB; fi
    HIP_PRECISION=$P python tools/stage_times.py 2>&1 | tail -1
  done
done
cd /tmp && export TMPDIR=/tmp
for L in "" 42; do
  if [ -n "$L" ]; then export VIPNERF_HIP_LIB=$ROOT/vip-nerf_amd/lib/libvipnerf_hip_exp$L.so; else unset VIPNERF_HIP_LIB; fi
  rm -rf /tmp/prof_pmc && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --precision fp16 --no-cpu-baseline --no-render --no-other-precisions > /tmp/prof_pmc.log 2>&1
  echo "WRITE_SIZE lib=exp$L"; python $ROOT/tools/pmc_summary.py /tmp/prof_pmc | grep "k_mlp" | cut -c1-140
done
