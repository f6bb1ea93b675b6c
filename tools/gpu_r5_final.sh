# round 5, final measurement set (v34: three tiers 31 -> 63 -> 127 rows, CM_FLAG_HFPRISM, the 40-dof model in two-wave form, sensors in
# front of J / outputs behind E on wave 1, chunk hand-over checks its XCD).  PART=1 suite + bench lines, PART=2 rocprofv3 kernel stats +
# stage stamps, PART=3 PMC passes + soak.  Box clocks differ between leases: the first bench line decides whether this box is a normal one.
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
PART=${PART:-1}
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    ws = d.get("workgroup_slots") or {}
    print(sys.argv[1].split("/")[-1], "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err %.1e" % d["max_qpos_err"], "kernel_ms %.3f stream_ms %.3f" % (d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step") or 0),
          "clock %.3f" % ((ws.get("clock_hz") or 0) / 1e9), "handed %.4f wide %.4f" % (d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch") or 0, d.get("frac_envs_in_the_127_row_pass_in_the_last_launch") or 0),
          {k: round(d[k]/1e6, 3) for k in ("value_exact_pd", "value_all_outputs_every_substep", "value_one_stream", "value_step_pd") if d.get(k)}, (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
B="--no-cpu-baseline --no-step-pd --no-other-mode"
if [ $PART = 1 ]; then
(rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|Perf" | head -8) > gpurun_out/box_clocks.txt
(time timeout 1800 python -m pytest tests -m gpu -q -s) > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
(time timeout 600 python bench.py 2> gpurun_out/bench_cassie.err | grep '^{"metric"' > gpurun_out/bench_cassie.json) 2> gpurun_out/bench_cassie.time
line gpurun_out/bench_cassie.json; tail -3 gpurun_out/bench_cassie.time
if ! python -c "import json,sys; sys.exit(0 if json.load(open('gpurun_out/bench_cassie.json'))['value'] >= 22.0e6 else 1)"; then echo SLOWBOX; cat gpurun_out/box_clocks.txt; exit 0; fi
(time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_cassie_short.err | grep '^{"metric"' > gpurun_out/bench_cassie_short.json) 2> gpurun_out/bench_cassie_short.time
line gpurun_out/bench_cassie_short.json; tail -3 gpurun_out/bench_cassie_short.time
CASSIE_WAVES_PER_ENV=1 timeout 300 python bench.py $B 2> gpurun_out/bench_cassie_one_wave.err | grep '^{"metric"' > gpurun_out/bench_cassie_one_wave.json; line gpurun_out/bench_cassie_one_wave.json
for m in cassie_hfield cassie_tray_box; do
  timeout 400 python bench.py --model $m --no-step-pd 2> gpurun_out/bench_$m.err | grep '^{"metric"' > gpurun_out/bench_$m.json; line gpurun_out/bench_$m.json
done
timeout 400 python bench.py --model cassie_hfield --hfield-contacts prism --no-step-pd 2> gpurun_out/bench_cassie_hfield_prism.err | grep '^{"metric"' > gpurun_out/bench_cassie_hfield_prism.json; line gpurun_out/bench_cassie_hfield_prism.json
CASSIE_TRAY_TWO_WAVES=0 timeout 300 python bench.py --model cassie_tray_box $B 2> gpurun_out/bench_tray_one_wave.err | grep '^{"metric"' > gpurun_out/bench_cassie_tray_box_one_wave.json; line gpurun_out/bench_cassie_tray_box_one_wave.json
for m in cassie cassie_hfield; do
  timeout 300 python bench.py --model $m --target-spread 10 $B 2> gpurun_out/bench_stress_$m.err | grep '^{"metric"' > gpurun_out/bench_stress_targets_$m.json; line gpurun_out/bench_stress_targets_$m.json
done
for m in cassie cassie_hfield; do timeout 200 python tools/handover_timing.py $m 2>&1 | tail -1; done | tee gpurun_out/handover_timing.txt
timeout 300 python bench.py --total-envs 65536 --steps 100 --warmup 50 --repeats 5 $B 2> gpurun_out/bench_total65536.err | grep '^{"metric"' > gpurun_out/bench_total65536.json; line gpurun_out/bench_total65536.json
timeout 300 python bench.py --envs-per-gpu 8192 --steps 200 --warmup 50 --force-collectives $B 2> gpurun_out/bench_8192_collectives.err | grep '^{"metric"' > gpurun_out/bench_8192_collectives.json; line gpurun_out/bench_8192_collectives.json
timeout 300 python tools/single_sim_profile.py > gpurun_out/single_sim_profile.txt 2>&1; tail -4 gpurun_out/single_sim_profile.txt
fi
R=$GRAFT_REPO_ROOT
if [ $PART = 2 ]; then
cd /tmp && export TMPDIR=/tmp
for m in cassie cassie_hfield cassie_tray_box; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python $R/bench.py --model $m $B > $R/gpurun_out/prof_$m.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie_short -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $B > $R/gpurun_out/prof_cassie_short.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie_hfield_prism -- python $R/bench.py --model cassie_hfield --hfield-contacts prism --steps 500 $B > $R/gpurun_out/prof_cassie_hfield_prism.log 2>&1
cd $R
for m in cassie cassie_hfield cassie_tray_box cassie_short cassie_hfield_prism; do f=$(ls -t gpurun_out/prof_$m/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_$m.csv; echo "== $m"; head -5 $f | cut -c1-200; done
NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_two_waves.txt 2>&1
NSUB=50 WAVES=1 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_one_wave.txt 2>&1
head -34 gpurun_out/stage_profile_nsub50_two_waves.txt
MODEL=cassie_hfield NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_hfield_two_waves.txt 2>&1
MODEL=cassie_tray_box NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_tray_two_waves.txt 2>&1
head -34 gpurun_out/stage_profile_nsub50_tray_two_waves.txt
fi
if [ $PART = 3 ]; then
for m in cassie cassie_tray_box cassie_hfield; do
  rm -rf gpurun_out/pmc; MODEL=$m bash tools/gpu_pmc_all.sh > gpurun_out/pmc_all_$m.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_$m.json
  python - <<PY
import json
d = json.load(open("gpurun_out/pmc_summary_$m.json"))["derived"]
print("$m", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != "note"})
PY
done
rm -rf gpurun_out/pmc
timeout 900 python bench.py --steps 10000 --warmup 100 --repeats 2 $B 2> gpurun_out/bench_soak.err | grep '^{"metric"' > gpurun_out/bench_soak_10000_steps_cassie.json; line gpurun_out/bench_soak_10000_steps_cassie.json
fi
