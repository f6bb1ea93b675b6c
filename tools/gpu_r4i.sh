#!/bin/bash
# round 4, call i: the hand-over list stays empty between launches in every mode (one wave / two waves per env)
mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_drive_parity_gpu.py -m gpu -x -q -k "two_wave_form or row_capped" > gpurun_out/r4i/tests.log 2>&1
echo "exit $?" >> gpurun_out/r4i/tests.log
tail -5 gpurun_out/r4i/tests.log
