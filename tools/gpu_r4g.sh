# round 4, seventh lease: the stages behind the solve (qacc, accelerometers, outputs, Euler) on wave 1
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log
ab() { # label env...
  lab=$1; shift
  for rep in 1 2; do
  env "$@" timeout 300 python bench.py --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4g.err | grep '^{"metric"' > gpurun_out/r4g_${lab}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4g_${lab}_$rep.json"))
print("$lab run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0)))
PY
  done
}
(ab two_waves CASSIE_WAVES_PER_ENV=2; ab one_wave CASSIE_WAVES_PER_ENV=1) 2>&1 | tee gpurun_out/ab_r4g.txt
NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_two_waves.txt 2>&1; cat gpurun_out/stage_profile_nsub50_two_waves.txt
for m in cassie_hfield; do
  for w in 2 1; do CASSIE_WAVES_PER_ENV=$w timeout 300 python bench.py --model $m --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4g_$m.err | grep '^{"metric"' > gpurun_out/r4g_$m.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4g_$m.json")); print("$m waves=$w: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"]))
PY
  done
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd 2> gpurun_out/r4g_short.err | grep '^{"metric"' > gpurun_out/r4g_short.json
python - <<PY
import json
d = json.load(open("gpurun_out/r4g_short.json")); print("driver command: %.3f M (min %.3f max %.3f) one-stream %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d.get("value_one_stream", 0)/1e6))
PY
python tools/single_sim_profile.py > gpurun_out/single_sim_profile.txt 2>&1; tail -4 gpurun_out/single_sim_profile.txt
