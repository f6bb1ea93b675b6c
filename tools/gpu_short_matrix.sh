#!/bin/bash
# the driver's command (--gpus 1 --steps 20 --warmup 5): streams x chunks per env-launch
mkdir -p gpurun_out
for rep in 1 2; do for st in 1 2 4; do for ch in 1 2 4; do
  CASSIE_CHUNKS=$ch timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $st --no-cpu-baseline --no-step-pd --no-other-mode 2>/dev/null | grep '^{"metric"' > gpurun_out/short_s${st}_c${ch}_$rep.json
  python - gpurun_out/short_s${st}_c${ch}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "%.3f M (%.2f .. %.2f)" % (d["value"]/1e6, d["value_min"]/1e6, d["value_max"]/1e6))
PY
done; done; done
