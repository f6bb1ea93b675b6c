mkdir -p gpurun_out
(python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log 2>&1
cat gpurun_out/pytest_gpu.log
