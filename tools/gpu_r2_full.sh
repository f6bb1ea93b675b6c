# the full measurement set of a kernel revision: GPU suite, bench of every config, short-run bench, rocprofv3 kernel stats,
# in-kernel stage profile, PMC passes
bash tools/gpu_r2a.sh > gpurun_out/r2a.log 2>&1
NSUB=50 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50.txt 2>&1
python tools/stage_profile.py 4096 > gpurun_out/stage_profile.txt 2>&1
bash tools/gpu_pmc_all.sh > gpurun_out/pmc_all.log 2>&1
MODE=exact-pd bash tools/gpu_pmc_hbm.sh > gpurun_out/pmc_hbm.log 2>&1
grep -E "passed|failed|smoke|cassie_sim_step_pd/s" gpurun_out/r2a.log; head -3 gpurun_out/stage_profile_nsub50.txt; tail -4 gpurun_out/pmc_hbm.log
