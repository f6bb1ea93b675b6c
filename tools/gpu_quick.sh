mkdir -p gpurun_out
(python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/pytest_gpu.log 2>&1
python tools/stage_profile.py 4096 > gpurun_out/stage_profile.txt 2>&1
cat gpurun_out/pytest_gpu.log gpurun_out/stage_profile.txt
