#!/bin/bash
# round 4: small batches (a single cassie_sim_t) through the two-wave full kernel with 512 registers a lane
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_drive_parity_gpu.py tests/test_dropin_gpu.py -m gpu -x -q -k "small_batch or dropin or sim" > gpurun_out/ab/tests.log 2>&1
echo "exit $?" >> gpurun_out/ab/tests.log; tail -3 gpurun_out/ab/tests.log
timeout 300 python tools/single_sim_profile.py > gpurun_out/ab/single_sim_profile.txt 2>&1; tail -4 gpurun_out/ab/single_sim_profile.txt
