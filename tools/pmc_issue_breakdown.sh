cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j8; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt; wc -l $O/sq_counters.txt
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pp && HIP_PRECISION=bf16 RAYS=32768 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pp -- python $R/tools/eval_time.py > /tmp/pp.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pp $O/eval_$T.txt | grep -i "mlp_fwd\|^kernel" | cut -c1-260 || tail -3 /tmp/pp.log
  rm -rf /tmp/pp && rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pp -- python $R/bench.py --steps 3 --warmup 1 --precision bf16 --no-cpu-baseline --no-render --no-other-precisions --no-configs4 --no-configs2 --no-sizes > /tmp/pp.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pp $O/train_$T.txt | grep -i "k_mlp\|^kernel" | cut -c1-260 || tail -3 /tmp/pp.log
done
