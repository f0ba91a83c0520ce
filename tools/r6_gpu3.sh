mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests/test_hip_round6.py tests/test_hip_step.py tests/test_hip_trajectory.py tests/test_hip_parity.py tests/test_hip_headline_size.py tests/test_hip_dist.py tests/test_hip_bf16.py -m gpu -q -x > gpurun_out/r6c/pytest.log 2>&1; tail -5 gpurun_out/r6c/pytest.log
for m in "fp32 module" "bf16 onecall"; do timeout 300 python tools/torch_ops_in_step.py $m > "gpurun_out/r6c/torch_ops_${m// /_}.log" 2>&1; done
head -40 gpurun_out/r6c/torch_ops_fp32_module.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6c/bench.json.log 2> gpurun_out/r6c/bench.err; tail -c 2500 gpurun_out/r6c/bench.json.log
cp gpurun_out/bench_full.json gpurun_out/r6c/bench_full.json
