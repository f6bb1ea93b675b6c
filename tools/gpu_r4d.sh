# round 4, fourth lease: the hand-over list (the pass behind the fast kernel as a small grid) -- tests, rate, kernel stats
mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_drive_parity_gpu.py tests/test_gpu_parity.py -m gpu -x -q -s) > gpurun_out/pytest_r4d.log 2>&1
tail -4 gpurun_out/pytest_r4d.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4d.err | grep '^{"metric"' > gpurun_out/r4d_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4d_$rep.json"))
print("hand-over list run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f stream_ms %.3f handed %s" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"], d["roofline"].get("stream_ms_per_policy_step", 0), d["frac_envs_handed_over_to_the_full_kernel_in_the_last_launch"]))
PY
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd 2> gpurun_out/r4d_short.err | grep '^{"metric"' > gpurun_out/r4d_short.json
python - <<PY
import json
d = json.load(open("gpurun_out/r4d_short.json")); print("driver command: %.3f M (min %.3f max %.3f) one-stream %.3f handed %s" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d.get("value_one_stream", 0)/1e6, d["frac_envs_handed_over_to_the_full_kernel_in_the_last_launch"]))
PY
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cassie -- python $R/bench.py --no-cpu-baseline --no-step-pd --no-other-mode > $R/gpurun_out/prof_cassie.log 2>&1
cd $R; f=$(ls -t gpurun_out/prof_cassie/*/*kernel_stats.csv | head -1); cp $f gpurun_out/kernel_stats_cassie.csv; head -6 $f | cut -c1-220
for m in cassie cassie_hfield; do WAVES=2 python tools/handover_timing.py $m; done
