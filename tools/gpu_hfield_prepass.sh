#!/bin/bash
# round 4: the height-field pre-pass with the wave sharing the cells under all sample spheres: parity, stage stamps, rate
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests -m gpu -x -q -k "hfield or terrain" > gpurun_out/ab/tests.log 2>&1
echo "exit $?" >> gpurun_out/ab/tests.log; tail -3 gpurun_out/ab/tests.log
MODEL=cassie_hfield NSUB=50 WAVES=2 timeout 300 python tools/stage_profile.py 4096 > gpurun_out/ab/stage_hfield_w2.txt 2>&1
sed -n 6,10p gpurun_out/ab/stage_hfield_w2.txt
for rep in 1 2; do
  timeout 400 python bench.py --model cassie_hfield --steps 500 --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode > gpurun_out/ab/bench_hfield_$rep.json 2> gpurun_out/ab/bench_hfield_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/bench_hfield_$rep.json").read().strip().split("\n")[-1])
w=d.get("workgroup_slots") or {}
print("hfield $rep: value %.3f M, kernel_ms %.3f stream_ms %.3f, handed %s, err %.2e, equal counts %s, slots busy %.3f (%.0f clocks per env-substep)" % (d["value"]/1e6, d["roofline"]["kernel_ms"], d["roofline"]["stream_ms_per_policy_step"], d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch"), d["parity"]["max_qpos_err"], d["parity"]["frac_envs_with_equal_ncon_nefc_iters"], w.get("busy_frac", 0), w.get("env_clocks_per_substep", 0)))
PY
done
