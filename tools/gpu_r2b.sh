mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
timeout 600 python bench.py > gpurun_out/bench_cassie.json 2> gpurun_out/bench_cassie.err
tail -15 gpurun_out/pytest_gpu.log; tail -c 2500 gpurun_out/bench_cassie.json; tail -3 gpurun_out/bench_cassie.err
