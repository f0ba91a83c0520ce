#!/bin/bash
# LDS-side PMC counters of the bench step (one counter per rocprofv3 pass, like tools/profile_round.sh): tools/profile_lds.sh [precision]
#   -> gpurun_out/round/pmc_<COUNTER>_<precision>.txt for SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_WAIT_INST_LDS, SQ_WAIT_INST_ANY, SQ_WAVE_CYCLES
PREC=${1:-bf16}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_WR; do
  rm -rf /tmp/prof_pmc && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --precision $PREC --no-cpu-baseline --no-render --no-other-precisions --no-configs4 > /tmp/prof_pmc.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/prof_pmc $OUT/pmc_${C}_$PREC.txt > /dev/null || tail -5 /tmp/prof_pmc.log
done
ls $OUT | grep "LDS\|WAIT\|WAVE_CYCLES\|VMEM_WR"
