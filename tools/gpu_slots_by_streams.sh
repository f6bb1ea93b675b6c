#!/bin/bash
# round 4: how full the workgroup slots are over the bench's timed regions (config 2, two streams / one / four)
mkdir -p gpurun_out/ab
for s in 2 1 4 3; do
  timeout 400 python bench.py --steps 400 --warmup 20 --streams $s --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/ab/bench_streams$s.json 2> gpurun_out/ab/bench_streams$s.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/bench_streams$s.json").read().strip().split("\n")[-1])
print("streams $s: value %.3f M, slots %s, kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, {k:(round(v,4) if isinstance(v,float) else v) for k,v in (d.get("workgroup_slots") or {}).items() if k!="note"}, d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"]))
PY
done
