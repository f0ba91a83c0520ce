cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for P in fp16x3 fp16 fp32; do
 for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS; do
  rm -rf /tmp/pp && HIP_PRECISION=$P RAYS=32768 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pp -- python $R/tools/eval_time.py > /tmp/pp.log 2>&1
  echo "== $P $C"; python $R/tools/pmc_summary.py /tmp/pp | grep -i "mlp_fwd" | cut -c1-200
 done
done
