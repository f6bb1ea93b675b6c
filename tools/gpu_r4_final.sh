# round 4, final measurement set (v32: two wavefronts per env for the Cassie instantiations -- mass-matrix group, drive-level pass,
# factorisations, bias / passive stage and the stages behind the solve on wave 1 --, hand-over list, list-walking two-wave pass).
# Box clocks differ by up to 30 % between leases: the first bench line decides whether this box is a normal one.
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
(rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|Perf" | head -8) > gpurun_out/box_clocks.txt
(time timeout 1800 python -m pytest tests -m gpu -q -s) > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err %.1e" % d["max_qpos_err"], "kernel_ms %.3f stream_ms %.3f" % (d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step") or 0),
          {k: round(d[k]/1e6, 3) for k in ("value_exact_pd", "value_all_outputs_every_substep", "value_one_stream", "value_step_pd") if d.get(k)}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
(time timeout 600 python bench.py 2> gpurun_out/bench_cassie.err | grep '^{"metric"' > gpurun_out/bench_cassie.json) 2> gpurun_out/bench_cassie.time
line gpurun_out/bench_cassie.json; tail -3 gpurun_out/bench_cassie.time
if ! python -c "import json,sys; sys.exit(0 if json.load(open('gpurun_out/bench_cassie.json'))['value'] >= 22.0e6 else 1)"; then echo SLOWBOX; cat gpurun_out/box_clocks.txt; exit 0; fi
(time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_cassie_short.err | grep '^{"metric"' > gpurun_out/bench_cassie_short.json) 2> gpurun_out/bench_cassie_short.time
line gpurun_out/bench_cassie_short.json; tail -3 gpurun_out/bench_cassie_short.time
CASSIE_WAVES_PER_ENV=1 timeout 300 python bench.py --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_cassie_one_wave.err | grep '^{"metric"' > gpurun_out/bench_cassie_one_wave.json; line gpurun_out/bench_cassie_one_wave.json
for m in cassie_hfield cassie_tray_box; do
  timeout 300 python bench.py --model $m --no-step-pd 2> gpurun_out/bench_$m.err | grep '^{"metric"' > gpurun_out/bench_$m.json; line gpurun_out/bench_$m.json
done
timeout 300 python bench.py --total-envs 65536 --steps 100 --warmup 50 --repeats 5 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_total65536.err | grep '^{"metric"' > gpurun_out/bench_total65536.json; line gpurun_out/bench_total65536.json
timeout 300 python bench.py --envs-per-gpu 8192 --steps 200 --warmup 50 --force-collectives --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_8192_collectives.err | grep '^{"metric"' > gpurun_out/bench_8192_collectives.json; line gpurun_out/bench_8192_collectives.json
timeout 300 python tools/single_sim_profile.py > gpurun_out/single_sim_profile.txt 2>&1; tail -4 gpurun_out/single_sim_profile.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in cassie cassie_hfield cassie_tray_box; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python $R/bench.py --model $m --no-cpu-baseline --no-step-pd --no-other-mode > $R/gpurun_out/prof_$m.log 2>&1
done
cd $R
for m in cassie cassie_hfield cassie_tray_box; do f=$(ls -t gpurun_out/prof_$m/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_$m.csv; echo "== $m"; head -4 $f | cut -c1-200; done
NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_two_waves.txt 2>&1
NSUB=50 WAVES=1 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_one_wave.txt 2>&1
head -34 gpurun_out/stage_profile_nsub50_two_waves.txt
for m in cassie cassie_tray_box cassie_hfield; do
  rm -rf gpurun_out/pmc; MODEL=$m bash tools/gpu_pmc_all.sh > gpurun_out/pmc_all_$m.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_$m.json
  python - <<PY
import json
d = json.load(open("gpurun_out/pmc_summary_$m.json"))["derived"]
print("$m", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != "note"})
PY
done
timeout 900 python bench.py --steps 10000 --warmup 100 --repeats 2 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_soak.err | grep '^{"metric"' > gpurun_out/bench_soak_10000_steps_cassie.json; line gpurun_out/bench_soak_10000_steps_cassie.json
for m in cassie_hfield cassie_tray_box; do
  timeout 900 python bench.py --model $m --steps 10000 --warmup 100 --repeats 2 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/bench_soak_$m.err | grep '^{"metric"' > gpurun_out/bench_soak_10000_steps_$m.json; line gpurun_out/bench_soak_10000_steps_$m.json
done
MODEL=cassie_hfield NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_hfield_two_waves.txt 2>&1
MODEL=cassie_tray_box NSUB=50 WAVES=1 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_tray_fast47.txt 2>&1
