mkdir -p gpurun_out/r6a; rm -f gpurun_out/r6a/grad.log
VIPNERF_GRAD_LOG=$PWD/gpurun_out/r6a/grad.log VIPNERF_GRAD_NOASSERT=1 timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r6a/pytest_gpu.log 2>&1; tail -5 gpurun_out/r6a/pytest_gpu.log
timeout 1200 python bench.py > gpurun_out/r6a/bench.json.log 2> gpurun_out/r6a/bench.err; tail -c 3000 gpurun_out/r6a/bench.json.log; tail -3 gpurun_out/r6a/bench.err
cp gpurun_out/bench_full.json gpurun_out/r6a/bench_full.json
