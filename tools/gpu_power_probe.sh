# Is the step kernel clock- / power-limited?  Samples socket power and the shader clock (rocm-smi) while bench.py runs the
# headline workload with one and with two waves per env; prints the samples next to the rates.
mkdir -p gpurun_out
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -E "Max|Perf" > gpurun_out/power_probe.txt
for w in 1 2; do
  (CASSIE_WAVES_PER_ENV=$w timeout 300 python bench.py --steps 4000 --warmup 100 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/power_w$w.err | grep '^{"metric"' > gpurun_out/power_w$w.json) &
  pid=$!
  sleep 6     # model load + pre-roll
  echo "== waves=$w" >> gpurun_out/power_probe.txt
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Socket|fclk|mclk" | tr '\n' ' ' >> gpurun_out/power_probe.txt; echo >> gpurun_out/power_probe.txt
    sleep 0.7
  done
  wait $pid
  python - <<PY >> gpurun_out/power_probe.txt
import json
d = json.load(open("gpurun_out/power_w$w.json")); print("waves=$w: %.3f M env-steps/s (min %.3f max %.3f) kernel_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["roofline"]["kernel_ms"]))
PY
done
cat gpurun_out/power_probe.txt
