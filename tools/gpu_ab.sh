#!/bin/bash
# tools/gpu_ab.sh VARIANT [MODELS] -- same-box A/B of the product library against cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_VARIANT.so
# (tools/build_variant.sh, or an older tree's build): bench.py alternating between the two, per model; prints one line per run.
# SUITE=1 runs the GPU suite on the product first; SHORT=1 adds the driver's command; STAGES=1 the two-wave stage stamps.
V=$1; MODELS=${2:-cassie}
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    ws = d.get("workgroup_slots") or {}
    print(sys.argv[1].split("/")[-1], "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err %.1e" % d["max_qpos_err"],
          "kernel_ms %.3f stream_ms %.3f" % (d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step") or 0),
          "clock %.3f GHz" % ((ws.get("clock_hz") or 0) / 1e9), "rows %.1f" % d["mean_constraint_rows"], "handed %.4f" % (d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch") or 0),
          {k: round(d[k]/1e6, 3) for k in ("value_exact_pd", "value_all_outputs_every_substep", "value_one_stream", "value_step_pd") if d.get(k)})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
if [ -n "$SUITE" ]; then
  (time timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
fi
VL=$PWD/cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_$V.so
for m in $MODELS; do
  for rep in 1 2; do
    for which in base new; do
      f=gpurun_out/ab_${m}_${which}_$rep.json
      if [ $which = base ]; then export CASSIE_LIB=$VL; else unset CASSIE_LIB; fi
      timeout 300 python bench.py --model $m ${STEPS:+--steps $STEPS} --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/ab_${m}_${which}_$rep.err | grep '^{"metric"' > $f; line $f
    done
  done
done
unset CASSIE_LIB
if [ -n "$SHORT" ]; then
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_short.err | grep '^{"metric"' > gpurun_out/bench_short.json; line gpurun_out/bench_short.json
fi
if [ -n "$STAGES" ]; then
  for m in $MODELS; do MODEL=$m NSUB=50 WAVES=2 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_${m}_two_waves.txt 2>&1; head -40 gpurun_out/stage_profile_nsub50_${m}_two_waves.txt; done
fi
