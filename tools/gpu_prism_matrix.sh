#!/bin/bash
# config 4 with CM_FLAG_HFPRISM: streams x (three tiers | fast -> 127-row pass alone) x floor of the passes' grids; and what a floor costs config 2
mkdir -p gpurun_out
run() { # label, env..., -- bench args
  local label=$1; shift
  env "$@" > /dev/null 2>&1
}
one() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py "$@" --no-cpu-baseline --no-step-pd --no-other-mode --steps 500 --repeats 4 2> gpurun_out/m_$label.err | grep '^{"metric"' > gpurun_out/m_$label.json
  python - gpurun_out/m_$label.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "%.3f M" % (d["value"]/1e6), "err %.1e" % d["max_qpos_err"], "kernel_ms %.2f stream_ms %.2f" % (d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step") or 0),
          "handed %.3f wide %.3f" % (d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch") or 0, d.get("frac_envs_in_the_127_row_pass_in_the_last_launch") or 0))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
P="--model cassie_hfield --hfield-contacts prism"
one prism_s2 X=1 -- $P --streams 2
one prism_s4 X=1 -- $P --streams 4
one prism_s8 X=1 -- $P --streams 8
one prism_s2_floor256 CASSIE_PASS_GRID_MIN=256 -- $P --streams 2
one prism_s4_floor256 CASSIE_PASS_GRID_MIN=256 -- $P --streams 4
one cassie_s2 X=1 -- --streams 2
one cassie_s2_floor256 CASSIE_PASS_GRID_MIN=256 -- --streams 2
one cassie_s2 X=1 -- --streams 2
one cassie_s2_floor256 CASSIE_PASS_GRID_MIN=256 -- --streams 2
