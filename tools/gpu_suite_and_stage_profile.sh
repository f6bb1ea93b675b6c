#!/bin/bash
# round 4: the whole GPU suite on the build with X as a pair of flags
mkdir -p gpurun_out/ab
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/ab/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/ab/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/ab/smoke.log 2>&1; tail -1 gpurun_out/ab/smoke.log
NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_two_waves.txt 2>&1; sed -n 6,32p gpurun_out/ab/stage_two_waves.txt
