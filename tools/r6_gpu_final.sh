# the final pass of round 6: smoke, the whole -m gpu suite, the default bench, end-to-end training runs through TrainerHip01 (module contract and one call), soak
mkdir -p gpurun_out/r6z; O=gpurun_out/r6z
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 1200 python bench.py > $O/bench.json.log 2> $O/bench.err; tail -c 700 $O/bench.json.log; cp gpurun_out/bench_full.json $O/bench_full.json
{ ONE_CALL=0 timeout 300 python tools/train_synthetic.py fp32 1500 64 1024; timeout 300 python tools/train_synthetic.py fp32 600 160 4096; timeout 300 python tools/train_synthetic.py bf16 3000 160 4096; } 2>&1 | grep -v amdgpu.ids > $O/train_long.log; cat $O/train_long.log | cut -c1-300
timeout 600 python tools/soak.py 800 2>&1 | grep -v amdgpu.ids | tail -5 > $O/soak.log; cat $O/soak.log | cut -c1-250
