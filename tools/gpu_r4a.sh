# round 4, first lease: the two-wave form of the fast kernels (phys_batch_set_waves_per_env) -- bit identity on the GPU,
# then A/B against the one-wave form on this box, alternating (box clocks differ by up to 30 % between leases).
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
(rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|Perf" | head -8) > gpurun_out/box_clocks.txt
(time timeout 900 python -m pytest tests/test_drive_parity_gpu.py -m gpu -x -q -s -k "two_wave or row_capped") > gpurun_out/pytest_two_wave.log 2>&1
tail -5 gpurun_out/pytest_two_wave.log
ab() { # model steps
for rep in 1 2; do for w in 1 2; do
  CASSIE_WAVES_PER_ENV=$w timeout 300 python bench.py --model $1 --steps $2 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/ab_$1_w$w.err | grep '^{"metric"' > gpurun_out/ab_$1_w${w}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_$1_w${w}_$rep.json"))
print("$1 waves=$w run $rep: %.3f M (min %.3f max %.3f) one-stream %.3f M err %.1e kernel_ms %.3f stream_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, (d.get("value_one_stream") or 0)/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0)))
PY
done; done
}
ab cassie 500 2>&1 | tee gpurun_out/occupancy_ab.txt
ab cassie_hfield 500 2>&1 | tee -a gpurun_out/occupancy_ab.txt
for w in 1 2; do CASSIE_WAVES_PER_ENV=$w timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd 2> gpurun_out/short_w$w.err | grep '^{"metric"' > gpurun_out/short_w$w.json
python - <<PY
import json
d = json.load(open("gpurun_out/short_w$w.json")); print("driver command, waves=$w: %.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6))
PY
done 2>&1 | tee -a gpurun_out/occupancy_ab.txt
