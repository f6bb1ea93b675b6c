#!/bin/bash
# round 4, call t: the whole GPU suite, smoke and the bench lines with launches in chunks (defaults: 4 whole-batch / 2 per range)
mkdir -p gpurun_out/r4t
(time timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/r4t/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r4t/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r4t/smoke.log 2>&1; tail -1 gpurun_out/r4t/smoke.log
show() { python - "$1" "$2" <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); w=d.get("workgroup_slots") or {}
print("%-24s %.3f M (min %.3f max %.3f) one stream %s exact %s all outputs %s step_pd %s kernel_ms %.3f stream_ms %.3f busy %.3f err %.1e" % (sys.argv[2], d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, *[("%.3f" % (d[k]/1e6)) if d.get(k) else "-" for k in ("value_one_stream","value_exact_pd","value_all_outputs_every_substep","value_step_pd")], d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"], w.get("busy_frac", 0), d["max_qpos_err"]))
PY
}
timeout 600 python bench.py 2> gpurun_out/r4t/bench_cassie.err | grep '^{"metric"' > gpurun_out/r4t/bench_cassie.json; show gpurun_out/r4t/bench_cassie.json "default"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4t/bench_cassie_short.err | grep '^{"metric"' > gpurun_out/r4t/bench_cassie_short.json; show gpurun_out/r4t/bench_cassie_short.json "driver's command"
for m in cassie_hfield cassie_tray_box; do
  timeout 300 python bench.py --model $m --no-step-pd --no-cpu-baseline 2> gpurun_out/r4t/bench_$m.err | grep '^{"metric"' > gpurun_out/r4t/bench_$m.json; show gpurun_out/r4t/bench_$m.json $m
done
timeout 300 python bench.py --total-envs 65536 --steps 100 --warmup 50 --repeats 5 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4t/bench_total65536.err | grep '^{"metric"' > gpurun_out/r4t/bench_total65536.json; show gpurun_out/r4t/bench_total65536.json "65536 in one batch"
timeout 300 python bench.py --envs-per-gpu 8192 --steps 200 --warmup 50 --force-collectives --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4t/bench_8192_collectives.err | grep '^{"metric"' > gpurun_out/r4t/bench_8192_collectives.json; show gpurun_out/r4t/bench_8192_collectives.json "8192 + collectives"
timeout 900 python bench.py --steps 10000 --warmup 100 --repeats 2 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4t/bench_soak.err | grep '^{"metric"' > gpurun_out/r4t/bench_soak_10000_steps_cassie.json; show gpurun_out/r4t/bench_soak_10000_steps_cassie.json "soak"
