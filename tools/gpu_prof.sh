mkdir -p gpurun_out
python tools/stage_profile.py 512 4096 > gpurun_out/stage_profile.txt 2>&1
cat gpurun_out/stage_profile.txt
