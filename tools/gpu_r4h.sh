# round 4, eighth lease: the 40-dof model with two waves per env (fast instantiation of 47 rows, lean Gram stage) against one wave
mkdir -p gpurun_out
(time timeout 1400 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log
for rep in 1 2; do for tw in 0 1; do
  CASSIE_TRAY_TWO_WAVES=$tw timeout 300 python bench.py --model cassie_tray_box --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4h_tray.err | grep '^{"metric"' > gpurun_out/r4h_tray_$tw.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4h_tray_$tw.json")); print("tray two_waves=$tw run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f rows %.1f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["mean_constraint_rows"]))
PY
done; done 2>&1 | tee gpurun_out/tray_two_waves_ab2.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4h.err | grep '^{"metric"' > gpurun_out/r4h_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4h_$rep.json"))
print("cassie --steps 1000 run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0)))
PY
done
