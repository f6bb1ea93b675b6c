#!/bin/bash
# round 4: the 40-dof model's 47-row one-wave instantiation with its Gram matrix on the matrix core (in the staged tile's
# own LDS): identity with the full kernel, parity with the oracle, and the rate against the full kernel alone (same box, alternating)
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_drive_parity_gpu.py tests/test_gpu_parity.py tests/test_config_parity_gpu.py -m gpu -x -q -k "tray or row_capped or two_wave" > gpurun_out/ab/tests.log 2>&1
echo "exit $?" >> gpurun_out/ab/tests.log
tail -4 gpurun_out/ab/tests.log
for rep in 1 2; do
for v in fast full; do
  if [ $v = full ]; then export CASSIE_NO_FAST_ROWS=1; else unset CASSIE_NO_FAST_ROWS; fi
  timeout 400 python bench.py --model cassie_tray_box --steps 500 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/ab/bench_tray_${v}_$rep.json 2> gpurun_out/ab/bench_tray_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/bench_tray_${v}_$rep.json").read().strip().split("\n")[-1])
print("$v $rep: value %.3f M, kernel_ms %.3f stream_ms %.3f, handed %s, parity %s" % (d["value"]/1e6, d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"], d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch"), d["parity"]["max_qpos_err"]))
PY
done
done
unset CASSIE_NO_FAST_ROWS
MODEL=cassie_tray_box NSUB=50 WAVES=1 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_tray_fast47.txt 2>&1
sed -n 6,22p gpurun_out/ab/stage_tray_fast47.txt
