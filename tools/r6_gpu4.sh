mkdir -p gpurun_out/r6d
# eval parity of the staggered kernel first (bf16 / fp16 eval goldens, free-running eval), then the A/B of the eval kernels, then the whole suite
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_hip_freerun.py tests/test_hip_round6.py tests/test_hip_step.py -m gpu -q -x > gpurun_out/r6d/pytest_eval.log 2>&1; tail -3 gpurun_out/r6d/pytest_eval.log
for rep in 1 2 3; do for L in "" nostag; do
  if [ -n "$L" ]; then export VIPNERF_HIP_LIB=$PWD/vip-nerf_amd/lib/libvipnerf_hip_$L.so; else unset VIPNERF_HIP_LIB; fi
  for P in bf16 fp16; do HIP_PRECISION=$P python tools/eval_time.py 2>&1 | tail -1; done
done; done > gpurun_out/r6d/ab_eval_stagger.log 2>&1
unset VIPNERF_HIP_LIB
cat gpurun_out/r6d/ab_eval_stagger.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6d/pytest_gpu.log 2>&1; tail -4 gpurun_out/r6d/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6d/bench.json.log 2> gpurun_out/r6d/bench.err; tail -c 1500 gpurun_out/r6d/bench.json.log
cp gpurun_out/bench_full.json gpurun_out/r6d/bench_full.json
