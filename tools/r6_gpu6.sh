# PMC view of the two 16-bit eval kernels (staggered product vs -DVN_PT2_EVAL_STAGGER=0): pipe-busy cycles, busy cycles (-> clock), LDS instructions, wait cycles
mkdir -p gpurun_out/r6f; cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for L in stag nostag; do
  if [ $L = nostag ]; then export VIPNERF_HIP_LIB=$ROOT/vip-nerf_amd/lib/libvipnerf_hip_nostag.so; else unset VIPNERF_HIP_LIB; fi
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_$L
    HIP_PRECISION=bf16 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$L -- python $ROOT/tools/eval_time.py > /tmp/pmc_$L.log 2>&1
    echo "== $L: $C" >> $ROOT/gpurun_out/r6f/pmc_eval.txt
    python $ROOT/tools/pmc_summary.py /tmp/pmc_$L 2>/dev/null | grep -E "^kernel|k_mlp" | cut -c1-260 >> $ROOT/gpurun_out/r6f/pmc_eval.txt
  done
done
cat $ROOT/gpurun_out/r6f/pmc_eval.txt
