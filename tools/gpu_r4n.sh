#!/bin/bash
# round 4, call n: the pass behind the 40-dof model's one-wave fast kernel as one-wave workgroups (default) or two-wave ones
mkdir -p gpurun_out/r4n
timeout 600 python -m pytest tests/test_drive_parity_gpu.py -m gpu -x -q -k "tray" > gpurun_out/r4n/tests.log 2>&1
echo "exit $?" >> gpurun_out/r4n/tests.log; tail -3 gpurun_out/r4n/tests.log
for rep in 1 2; do
for v in pass1w pass2w; do
  if [ $v = pass2w ]; then export CASSIE_DEBUG_TRAY_PASS_TWO_WAVES=1; else unset CASSIE_DEBUG_TRAY_PASS_TWO_WAVES; fi
  timeout 400 python bench.py --model cassie_tray_box --steps 500 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/r4n/bench_tray_${v}_$rep.json 2> gpurun_out/r4n/bench_tray_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4n/bench_tray_${v}_$rep.json").read().strip().split("\n")[-1])
w=d.get("workgroup_slots") or {}
print("$v $rep: value %.3f M, kernel_ms %.3f stream_ms %.3f, handed %s, slots busy %.3f (%.0f clocks per env-substep)" % (d["value"]/1e6, d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"], d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch"), w.get("busy_frac", 0), w.get("env_clocks_per_substep", 0)))
PY
done
done
