#!/bin/bash
# round 4: does the launch order (expensive envs of the last launch first) still pay?  CASSIE_NO_BALANCE=1 against the default
mkdir -p gpurun_out/ab
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); w=d.get("workgroup_slots") or {}
print("%-34s %.3f M (min %.3f max %.3f) one stream %s kernel_ms %.3f stream_ms %.3f" % (sys.argv[2], d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, ("%.3f" % (d["value_one_stream"]/1e6)) if d.get("value_one_stream") else "-", d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"]))
PY
}
for rep in 1 2; do
for m in cassie cassie_tray_box; do
for v in order noorder; do
  if [ $v = noorder ]; then export CASSIE_NO_BALANCE=1; else unset CASSIE_NO_BALANCE; fi
  timeout 300 python bench.py --model $m --steps 500 --warmup 50 --no-cpu-baseline --no-step-pd > gpurun_out/ab/${m}_${v}_$rep.json 2> gpurun_out/ab/${m}_${v}_$rep.err; show gpurun_out/ab/${m}_${v}_$rep.json "$m $v run $rep"
done
done
done
