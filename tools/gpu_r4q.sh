#!/bin/bash
mkdir -p gpurun_out/r4q
MODEL=cassie_hfield NSUB=50 WAVES=2 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/r4q/stage_hfield_w2.txt 2>&1
grep -E "pre-pass|collision|TWO WAVES" gpurun_out/r4q/stage_hfield_w2.txt
