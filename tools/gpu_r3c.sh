# round 3: the row-capped fast kernel -- GPU suite, then A/B bench lines (fast kernel on / off) of cassie and cassie_hfield
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
for rep in 1 2; do
for m in cassie cassie_hfield; do
  for ab in fast full; do
    if [ $ab = full ]; then export CASSIE_NO_FAST_ROWS=1; else unset CASSIE_NO_FAST_ROWS; fi
    timeout 300 python bench.py --model $m --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_${m}_$ab.err | grep '^{"metric"' > gpurun_out/bench_${m}_$ab.json
    python - <<PY
import json
d = json.load(open("gpurun_out/bench_${m}_$ab.json"))
print("$m $ab", "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err", d["max_qpos_err"], "kernel_ms", d["roofline"]["kernel_ms"], "rows", d["mean_constraint_rows"])
PY
  done
done
done
unset CASSIE_NO_FAST_ROWS
