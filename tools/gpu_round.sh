#!/bin/bash
# The round's validation pass on a GPU box: the whole -m gpu suite, the default bench line, then rocprofv3 kernel stats + PMC passes
# (fp32, bf16).  Output under gpurun_out/round/.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/round; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 1200 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err; tail -c 600 $OUT/bench.json.log; tail -3 $OUT/bench.err
cp gpurun_out/bench_full.json $OUT/bench_full.json      # (the profiling passes below overwrite gpurun_out/bench_full.json with their own short runs)
SKIP_TESTS=1 bash tools/profile_round.sh fp32 > $OUT/prof_fp32.log 2>&1
SKIP_TESTS=1 bash tools/profile_round.sh bf16 > $OUT/prof_bf16.log 2>&1
ls $OUT | wc -l
