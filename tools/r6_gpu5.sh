mkdir -p gpurun_out/r6e
timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_freerun.py -m gpu -q -x -k "eval or bf16 or render" > gpurun_out/r6e/pytest_eval.log 2>&1; tail -3 gpurun_out/r6e/pytest_eval.log
for rep in 1 2 3; do for L in "" nostag; do
  if [ -n "$L" ]; then export VIPNERF_HIP_LIB=$PWD/vip-nerf_amd/lib/libvipnerf_hip_$L.so; else unset VIPNERF_HIP_LIB; fi
  HIP_PRECISION=bf16 python tools/eval_time.py 2>&1 | tail -1
done; done > gpurun_out/r6e/ab_eval_stagger.log 2>&1
cat gpurun_out/r6e/ab_eval_stagger.log
