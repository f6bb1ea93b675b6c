# round 4, sixth lease: compiler scheduling variants on the two-wave kernels; the 40-dof kernel with two waves per env
mkdir -p gpurun_out
VARIANTS="tree defsched maxilp licm" STEPS=500 bash tools/gpu_variants.sh 2>&1 | tee gpurun_out/variants_ab_two_wave.txt
for rep in 1 2; do for tw in 0 1; do
  if [ $tw = 1 ]; then export CASSIE_TRAY_TWO_WAVES=1 CASSIE_WAVES_PER_ENV=2; else unset CASSIE_TRAY_TWO_WAVES; export CASSIE_WAVES_PER_ENV=2; fi
  timeout 300 python bench.py --model cassie_tray_box --steps 500 --warmup 50 --repeats 6 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/r4f_tray.err | grep '^{"metric"' > gpurun_out/r4f_tray_$tw.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r4f_tray_$tw.json")); print("tray two_waves=$tw run $rep: %.3f M (min %.3f max %.3f) err %.1e kernel_ms %.3f" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6, d["max_qpos_err"], d["roofline"]["kernel_ms"]))
PY
done; done 2>&1 | tee gpurun_out/tray_two_waves_ab.txt
