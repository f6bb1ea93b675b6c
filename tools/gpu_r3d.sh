# round 3: the v22 measurement set -- GPU suite, launch-boundary cost curve (substeps per launch 1..50, the lower bound of what
# any per-substep kernel split would pay), rocprofv3 kernel stats of the three configs, PMC passes of the headline kernel,
# in-kernel stage profile (full kernel: the stamps need io.prof)
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -x -q -s) > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed|handed over" gpurun_out/pytest_gpu.log
echo "substeps_per_launch value_M kernel_ms_per_launch ms_per_step" > gpurun_out/boundary_curve.txt
for spl in 50 25 10 5 2 1; do
  timeout 300 python bench.py --substeps-per-launch $spl --steps 200 --warmup 50 --repeats 5 --no-cpu-baseline --no-step-pd --no-other-mode 2> /dev/null | grep '^{"metric"' > gpurun_out/bench_spl$spl.json
  python - <<PY >> gpurun_out/boundary_curve.txt
import json
d = json.load(open("gpurun_out/bench_spl$spl.json"))
print($spl, "%.3f" % (d["value"]/1e6), "%.4f" % d["roofline"]["kernel_ms"], "%.4f" % d["ms_per_step"])
PY
done
cat gpurun_out/boundary_curve.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in cassie cassie_hfield cassie_tray_box; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python $R/bench.py --model $m --steps 200 --warmup 50 --repeats 3 --no-cpu-baseline --no-step-pd --no-other-mode > $R/gpurun_out/prof_$m.log 2>&1
done
cd $R
for m in cassie cassie_hfield cassie_tray_box; do f=$(ls gpurun_out/prof_$m/*/*kernel_stats.csv | tail -1); echo "== $m"; head -6 $f; done
NSUB=50 python tools/stage_profile.py 4096 > gpurun_out/stage_profile_nsub50.txt 2>&1
head -20 gpurun_out/stage_profile_nsub50.txt
bash tools/gpu_pmc_all.sh > gpurun_out/pmc_all.log 2>&1
tail -30 gpurun_out/pmc_summary.json
