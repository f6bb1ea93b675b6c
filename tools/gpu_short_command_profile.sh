#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats of the driver's command (bench.py --gpus 1 --steps 20 --warmup 5, device legs only) beside the
# line's roofline.kernel_ms
mkdir -p gpurun_out/ab
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode 2> gpurun_out/ab/short.err | grep '^{"metric"' > gpurun_out/ab/bench_cassie_short_device_legs.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab/prof_short -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-step-pd --no-other-mode > $R/gpurun_out/ab/prof_short.log 2>&1
cd $R
f=$(ls -t gpurun_out/ab/prof_short/*/*kernel_stats.csv | head -1); cp $f gpurun_out/ab/kernel_stats_cassie_short.csv; head -4 $f | cut -c1-160
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab/bench_cassie_short_device_legs.json")); print("line: value %.3f M kernel_ms %.4f launches %d" % (d["value"]/1e6, d["roofline"]["kernel_ms"], d["roofline"]["kernel_launches_timed"]))
PY
