# range stepping on two streams: GPU suite, then the bench line of every config (with the one-stream leg for comparison)
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-step-pd 2> gpurun_out/bench_cassie.err | grep '^{"metric"' > gpurun_out/bench_cassie.json
for m in cassie_hfield cassie_tray_box; do
  timeout 300 python bench.py --model $m --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_$m.err | grep '^{"metric"' > gpurun_out/bench_$m.json
  timeout 300 python bench.py --model $m --streams 1 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_${m}_1s.err | grep '^{"metric"' > gpurun_out/bench_${m}_1s.json
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd 2> gpurun_out/bench_cassie_short.err | grep '^{"metric"' > gpurun_out/bench_cassie_short.json
timeout 300 python bench.py --envs-per-gpu 8192 --steps 200 --warmup 50 --force-collectives --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_8192_collectives.err | grep '^{"metric"' > gpurun_out/bench_8192_collectives.json
python - <<'PY'
import json
for f in ("cassie", "cassie_short", "cassie_hfield", "cassie_hfield_1s", "cassie_tray_box", "cassie_tray_box_1s", "8192_collectives"):
    try:
        d = json.load(open("gpurun_out/bench_%s.json" % f))
        print("%-20s %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f streams %s" % (f, d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["config"]["streams"]),
              {k: round(d[k]/1e6, 3) for k in ("value_exact_pd", "value_all_outputs_every_substep", "value_one_stream") if d.get(k)}, d["parity"]["frac_envs_with_equal_ncon_nefc_iters"], d.get("obs_allgather_ok"))
    except Exception as e:
        print(f, "no line:", e)
PY
tail -3 gpurun_out/bench_cassie.err
