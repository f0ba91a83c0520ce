#!/bin/bash
# tools/lds_ratio_probe with socket power / shader clock sampled twice a second next to it (on the GPU box)
mkdir -p gpurun_out/r6_probe
( while true; do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Socket Power|power' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.5; done ) > gpurun_out/r6_probe/smi.log &
SMI=$!
tools/bin/lds_ratio_probe 4 2>&1 | while IFS= read -r line; do echo "$(date +%s.%N | cut -c1-14) $line"; done > gpurun_out/r6_probe/probe.log
kill $SMI
cat gpurun_out/r6_probe/probe.log
