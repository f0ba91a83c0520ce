#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round (run through gpurun from the repo root):
#   tools/profile_round.sh [precision]      -> gpurun_out/round/{pytest_gpu.log,bench.json.log,kernel_stats.txt,pmc_<COUNTER>.txt}
# rocprofv3: --kernel-trace --stats for the per-kernel times; the PMC counters in separate passes with --kernel-trace only.
PREC=${1:-fp32}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/round
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json.log 2> $OUT/bench.err; tail -c 600 $OUT/bench.json.log
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks && rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -- python $ROOT/bench.py --steps 5 --warmup 2 --precision $PREC --no-cpu-baseline --no-render --no-other-precisions --no-configs4 --no-configs2 --no-sizes --repeats 1 > /tmp/prof_ks.log 2>&1
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
if [ -n "$DB" ]; then python $ROOT/tools/rocpd_summary.py $DB $OUT/kernel_stats_$PREC.txt > /dev/null; else find /tmp/prof_ks -name '*stats*' | head; tail -5 /tmp/prof_ks.log; fi
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/prof_pmc && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --precision $PREC --no-cpu-baseline --no-render --no-other-precisions --no-configs4 --no-configs2 --no-sizes --repeats 1 > /tmp/prof_pmc.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/prof_pmc $OUT/pmc_${C}_$PREC.txt > /dev/null || tail -5 /tmp/prof_pmc.log
done
ls -la $OUT
