# round 4, third lease: the pass behind the fast kernel as two-wave workgroups (placeable wherever a fast workgroup is) against
# the one-wave full kernel (needs an empty SIMD) and against no pass at all; per-env launch cost; hand-over-heavy workload.
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_drive_parity_gpu.py -m gpu -x -q -s -k "two_wave or row_capped or launch_order") > gpurun_out/pytest_r4c.log 2>&1
tail -4 gpurun_out/pytest_r4c.log
ab() { # label env...
  lab=$1; shift
  for rep in 1 2; do
  env "$@" timeout 300 python bench.py --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4c.err | grep '^{"metric"' > gpurun_out/r4c_${lab}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4c_${lab}_$rep.json"))
print("$lab run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0)))
PY
  done
}
(ab two_wave_pass CASSIE_WAVES_PER_ENV=2; ab one_wave_pass CASSIE_WAVES_PER_ENV=2 CASSIE_DEBUG_RESUME_ONE_WAVE=1; ab no_pass CASSIE_WAVES_PER_ENV=2 CASSIE_DEBUG_SKIP_RESUME_PASS=1; ab one_wave_everything CASSIE_WAVES_PER_ENV=1) 2>&1 | tee gpurun_out/resume_pass_ab2.txt
for w in 2 1; do NSUB=50 WAVES=$w python tools/stage_profile.py 4096 2>&1 | grep -E "nenv|whole launch"; NSUB=50 WAVES=$w TWO_STREAM_LOAD=1 python tools/stage_profile.py 4096 2>&1 | grep -E "nenv|whole launch"; done | tee gpurun_out/launch_cost.txt
for m in cassie cassie_hfield; do
  WAVES=2 python tools/handover_timing.py $m; WAVES=2 CASSIE_DEBUG_RESUME_ONE_WAVE=1 python tools/handover_timing.py $m; WAVES=1 python tools/handover_timing.py $m
done 2>&1 | tee gpurun_out/handover_timing.txt
