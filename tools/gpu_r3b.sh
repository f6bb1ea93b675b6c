# round 3: quick A/B of a kernel revision -- GPU suite, the three configs' bench lines (device legs only), single-sim profile
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
for m in cassie cassie_hfield cassie_tray_box; do
  timeout 300 python bench.py --model $m --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_$m.err | grep '^{"metric"' > gpurun_out/bench_$m.json
done
timeout 300 python tools/single_sim_profile.py > gpurun_out/single_sim_profile.txt 2>&1
tail -4 gpurun_out/pytest_gpu.log
for m in cassie cassie_hfield cassie_tray_box; do python - <<PY
import json
d = json.load(open("gpurun_out/bench_$m.json"))
print("$m", "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err", d["max_qpos_err"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
done
cat gpurun_out/single_sim_profile.txt
