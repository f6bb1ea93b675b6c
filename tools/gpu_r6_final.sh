# round 6, final measurement set (v36; v37 = the same with the fast kernel's form picked per env range: per-env parameter blocks + device set_const, CM_DRIVE_PD_SAFE as the benchmarked mode, tighter
# hand-over verdict, CM_FLAG_BOX8, one-wave form for large alone launches).  PART=1 suite + bench lines, PART=2 rocprofv3 kernel stats +
# stage stamps + resources, PART=3 PMC passes + soak.  Box clocks differ between leases: the first bench line says what this box is.
mkdir -p gpurun_out; nproc > gpurun_out/nproc.txt
PART=${PART:-1}
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    ws = d.get("workgroup_slots") or {}
    print(sys.argv[1].split("/")[-1], "%.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6), "err %.1e" % d["max_qpos_err"], "mode", d["config"]["mode"],
          "kernel_ms %.3f stream_ms %.3f" % (d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step") or 0), "clock %.3f" % ((ws.get("clock_hz") or 0) / 1e9),
          "handed %.4f" % (d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch") or 0),
          {k: round(d[k]/1e6, 3) for k in d if k.startswith("value_") and k not in ("value_min", "value_max") and d.get(k)}, (d.get("cpu_baseline") or {}).get("value"))
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
(time timeout 900 python bench.py 2> gpurun_out/bench_cassie.err | grep '^{"metric"' > gpurun_out/bench_cassie.json) 2> gpurun_out/bench_cassie.time
line gpurun_out/bench_cassie.json; tail -3 gpurun_out/bench_cassie.time
(time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_cassie_short.err | grep '^{"metric"' > gpurun_out/bench_cassie_short.json) 2> gpurun_out/bench_cassie_short.time
line gpurun_out/bench_cassie_short.json; tail -3 gpurun_out/bench_cassie_short.time
timeout 300 python bench.py --mode drive-pd $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_cassie_drive_pd.json; line gpurun_out/bench_cassie_drive_pd.json
timeout 300 python bench.py --randomise 4242 $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_cassie_randomised.json; line gpurun_out/bench_cassie_randomised.json
CASSIE_WAVES_PER_ENV=1 timeout 300 python bench.py $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_cassie_one_wave.json; line gpurun_out/bench_cassie_one_wave.json
for m in cassie_hfield cassie_tray_box; do
  timeout 400 python bench.py --model $m --no-step-pd 2> gpurun_out/bench_$m.err | grep '^{"metric"' > gpurun_out/bench_$m.json; line gpurun_out/bench_$m.json
done
timeout 400 python bench.py --model cassie_tray_box --box-contacts 8 $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_cassie_tray_box_box8.json; line gpurun_out/bench_cassie_tray_box_box8.json
timeout 400 python bench.py --model cassie_hfield --hfield-contacts prism $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_cassie_hfield_prism.json; line gpurun_out/bench_cassie_hfield_prism.json
for m in cassie cassie_hfield; do
  timeout 300 python bench.py --model $m --target-spread 10 $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_stress_targets_$m.json; line gpurun_out/bench_stress_targets_$m.json
done
timeout 300 python bench.py --total-envs 65536 --steps 100 --warmup 50 --repeats 5 $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_total65536.json; line gpurun_out/bench_total65536.json
timeout 300 python bench.py --envs-per-gpu 8192 --steps 200 --warmup 50 --force-collectives $B 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_8192_collectives.json; line gpurun_out/bench_8192_collectives.json
timeout 300 python tools/single_sim_profile.py > gpurun_out/single_sim_profile.txt 2>&1; tail -4 gpurun_out/single_sim_profile.txt
fi
R=$GRAFT_REPO_ROOT
if [ $PART = 2 ]; then
cd /tmp && export TMPDIR=/tmp
for m in cassie cassie_hfield cassie_tray_box; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python $R/bench.py --model $m $B > $R/gpurun_out/prof_$m.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie_short -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $B > $R/gpurun_out/prof_cassie_short.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie_randomised -- python $R/bench.py --randomise 4242 $B > $R/gpurun_out/prof_cassie_randomised.log 2>&1
cd $R
for m in cassie cassie_hfield cassie_tray_box cassie_short cassie_randomised; do f=$(ls -t gpurun_out/prof_$m/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_$m.csv; echo "== $m"; head -6 $f | cut -c1-220; done
NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_two_waves.txt 2>&1
head -34 gpurun_out/stage_profile_nsub50_two_waves.txt
MODEL=cassie_tray_box NSUB=50 WAVES=2 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50_tray_two_waves.txt 2>&1
fi
if [ $PART = 3 ]; then
for m in cassie cassie_tray_box; do
  rm -rf gpurun_out/pmc; MODEL=$m MODE=drive-pd-safe bash tools/gpu_pmc_all.sh > gpurun_out/pmc_all_$m.log 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/pmc_summary_$m.json
  python - <<PY
import json
d = json.load(open("gpurun_out/pmc_summary_$m.json"))["derived"]
print("$m", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != "note"})
PY
done
rm -rf gpurun_out/pmc
timeout 900 python bench.py --steps 10000 --warmup 100 --repeats 2 $B 2> gpurun_out/bench_soak.err | grep '^{"metric"' > gpurun_out/bench_soak_10000_steps_cassie.json; line gpurun_out/bench_soak_10000_steps_cassie.json
fi
