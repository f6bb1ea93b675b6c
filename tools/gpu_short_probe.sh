for rep in 1 2; do for spl in 50 10 5; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --substeps-per-launch $spl --no-cpu-baseline --no-step-pd --no-other-mode 2>/dev/null | grep '^{"metric"' > gpurun_out/short_$spl.json
  python - <<PY
import json
d=json.load(open("gpurun_out/short_$spl.json")); print("steps 20, substeps/launch $spl: %.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6))
PY
done; done
for spl in 50 10; do
  timeout 300 python bench.py --substeps-per-launch $spl --no-cpu-baseline --no-step-pd --no-other-mode 2>/dev/null | grep '^{"metric"' > gpurun_out/long_$spl.json
  python - <<PY
import json
d=json.load(open("gpurun_out/long_$spl.json")); print("steps 1000, substeps/launch $spl: %.3f M (min %.3f max %.3f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6))
PY
done
