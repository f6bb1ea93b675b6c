#!/bin/bash
# round 4: stepping launches in chunks (phys_batch_set_chunks / CASSIE_CHUNKS): identity, and the rate by chunk count --
# long regions (two streams / one stream) and the driver's short command
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_drive_parity_gpu.py -m gpu -x -q -s -k "chunks" > gpurun_out/ab/tests.log 2>&1
echo "exit $?" >> gpurun_out/ab/tests.log; grep -E "handed over|passed|failed|Error|error|exit" gpurun_out/ab/tests.log | tail -8
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); w=d.get("workgroup_slots") or {}
print("%-28s %.3f M (min %.3f max %.3f) one stream %s  kernel_ms %.3f stream_ms %.3f  slots busy %.3f  err %.1e" % (sys.argv[2], d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, ("%.3f" % (d["value_one_stream"]/1e6)) if d.get("value_one_stream") else "-", d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"], w.get("busy_frac", 0), d["max_qpos_err"]))
PY
}
for rep in 1 2; do
for k in 1 default; do
  if [ $k = default ]; then unset CASSIE_CHUNKS; else export CASSIE_CHUNKS=$k; fi
  timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-step-pd > gpurun_out/ab/long_k${k}_$rep.json 2> gpurun_out/ab/long_k${k}_$rep.err; show gpurun_out/ab/long_k${k}_$rep.json "long chunks=$k run $rep"
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/ab/short_k${k}_$rep.json 2> gpurun_out/ab/short_k${k}_$rep.err; show gpurun_out/ab/short_k${k}_$rep.json "short chunks=$k run $rep"
done
done
unset CASSIE_CHUNKS
