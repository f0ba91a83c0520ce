#!/bin/bash
# The test's own command (bench.py spawning 8 ranks on ONE GPU over gloo), N times, with the HIP runtime's warnings and errors on
# (AMD_LOG_LEVEL=2): a start in which a rank dies by a signal keeps its whole stderr under gpurun_out/spawn8/.
N=${1:-20}; mkdir -p gpurun_out/spawn8; fails=0
for i in $(seq 1 $N); do
  VIPNERF_DIST_BACKEND=gloo AMD_LOG_LEVEL=2 TORCH_DISTRIBUTED_DEBUG=OFF python bench.py --gpus 8 --steps 1 --warmup 0 --scaling strong --global-rays 8192 \
      --no-configs4 --no-configs2 --no-render --no-sizes --also "" > /tmp/sb.out 2> /tmp/sb.err; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp /tmp/sb.err gpurun_out/spawn8/bench_start_${i}_rc${rc}.err; echo "start $i: rc $rc"; grep -n "Signal\|abort\|terminate\|what()\|HSA\|hipError\|error\|Error" /tmp/sb.err | head -30; fi
done
echo "$N starts of 8 ranks: $fails failed"
