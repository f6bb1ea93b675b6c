# round 4, fifth lease: drive-level pass + bias / passive stage on wave 1 (LDS flag), hand-over list as a template parameter
mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests/test_drive_parity_gpu.py tests/test_gpu_parity.py tests/test_drive_io_gpu.py tests/test_config_parity_gpu.py -m gpu -x -q) > gpurun_out/pytest_r4e.log 2>&1
tail -4 gpurun_out/pytest_r4e.log
ab() { # label env...
  lab=$1; shift
  for rep in 1 2; do
  env "$@" timeout 300 python bench.py --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4e.err | grep '^{"metric"' > gpurun_out/r4e_${lab}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4e_${lab}_$rep.json"))
print("$lab run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0)))
PY
  done
}
(ab two_waves CASSIE_WAVES_PER_ENV=2; ab one_wave CASSIE_WAVES_PER_ENV=1; ab hfield_two_waves CASSIE_WAVES_PER_ENV=2 BENCH_MODEL=cassie_hfield) 2>&1 | tee gpurun_out/ab_r4e.txt
for w in 2; do NSUB=50 WAVES=$w python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_two_waves.txt 2>&1; cat gpurun_out/stage_profile_nsub50_two_waves.txt; done
for m in cassie_hfield cassie_tray_box; do
  timeout 300 python bench.py --model $m --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4e_$m.err | grep '^{"metric"' > gpurun_out/r4e_$m.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4e_$m.json")); print("$m: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"]))
PY
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd 2> gpurun_out/r4e_short.err | grep '^{"metric"' > gpurun_out/r4e_short.json
python - <<PY
import json
d = json.load(open("gpurun_out/r4e_short.json")); print("driver command: %.3f M (min %.3f max %.3f) one-stream %.3f handed %s" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d.get("value_one_stream", 0)/1e6, d["frac_envs_handed_over_to_the_full_kernel_in_the_last_launch"]))
PY
