#!/bin/bash
# round 5: the GPU suite on the three-tier build, config 4 with both height-field contact definitions, the single simulator
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
(time timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err %.1e" % d["max_qpos_err"],
          "kernel_ms %.3f stream_ms %.3f" % (d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step") or 0),
          "rows %.1f sweeps %.1f" % (d["mean_constraint_rows"], d["mean_pgs_iterations"]), "handed %.4f wide %.4f warn %d" % (d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch") or 0, d.get("frac_envs_in_the_127_row_pass_in_the_last_launch") or 0, d.get("envs_with_warnings") or 0),
          (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
for c in default prism; do
  timeout 400 python bench.py --model cassie_hfield --hfield-contacts $c --no-step-pd 2> gpurun_out/bench_hfield_$c.err | grep '^{"metric"' > gpurun_out/bench_hfield_$c.json; line gpurun_out/bench_hfield_$c.json
done
timeout 300 python tools/single_sim_profile.py > gpurun_out/single_sim_profile.txt 2>&1; tail -6 gpurun_out/single_sim_profile.txt
