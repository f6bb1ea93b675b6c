#!/bin/bash
# round 4: stage stamps of the height-field model (two waves / one wave per env)
mkdir -p gpurun_out/ab
MODEL=cassie_hfield NSUB=50 WAVES=2 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_hfield_w2.txt 2>&1
MODEL=cassie_hfield NSUB=50 WAVES=1 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_hfield_w1.txt 2>&1
cat gpurun_out/ab/stage_hfield_w2.txt | sed -n 6,36p
