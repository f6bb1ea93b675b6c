#!/bin/bash
# round 4: stage stamps of the 40-dof model (cassie_tray_box.xml), one wave per env
mkdir -p gpurun_out/ab
MODEL=cassie_tray_box NSUB=50 WAVES=1 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_tray_w1.txt 2>&1
MODEL=cassie_tray_box NSUB=50 WAVES=2 FULL_KERNEL=1 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_tray_w2_full.txt 2>&1
cat gpurun_out/ab/stage_tray_w1.txt
