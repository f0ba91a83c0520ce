#!/bin/bash
# The round's validation pass on a GPU box: the whole -m gpu suite, then the default bench line.  Output under gpurun_out/round/.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/round; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 1200 python bench.py > $OUT/bench.json.log 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json.log; tail -5 $OUT/bench.err
