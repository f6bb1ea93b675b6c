#!/bin/bash
# round 4: the clock the chip runs at under the step kernel (shader clock against the 100 MHz wall clock), and how full
# the workgroup slots are over a launch
mkdir -p gpurun_out/clock_and_slots
export DUMP=gpurun_out/clock_and_slots/raw
for w in 2 1; do
  NSUB=50 WAVES=$w timeout 300 python tools/stage_profile.py 4096 > gpurun_out/clock_and_slots/stage_w$w.txt 2>&1
  NSUB=50 WAVES=$w TWO_STREAM_LOAD=1 timeout 300 python tools/stage_profile.py 2048 > gpurun_out/clock_and_slots/stage_w${w}_two_stream_2048.txt 2>&1
done
NSUB=50 WAVES=2 timeout 300 python tools/stage_profile.py 1024 8192 > gpurun_out/clock_and_slots/stage_w2_1024_8192.txt 2>&1
head -8 gpurun_out/clock_and_slots/stage_w2.txt; head -8 gpurun_out/clock_and_slots/stage_w1.txt
