#!/bin/bash
# config 5, alternating on one box: one wave per env against the two-wave instantiations (CASSIE_TRAY_TWO_WAVES=1), the latter with A in
# LDS through the sweeps (the product) and with the lane's row of A in registers (variant `trayregs`, tools/build_variant.sh)
mkdir -p gpurun_out
VL=$PWD/cassie-mujoco-sim_amd/lib/variants/libcassiemujoco_trayregs.so
one() { # label, waves, lib
  if [ -n "$3" ]; then export CASSIE_LIB=$3; else unset CASSIE_LIB; fi
  CASSIE_TRAY_TWO_WAVES=$2 timeout 300 python bench.py --model cassie_tray_box --no-cpu-baseline --no-step-pd --no-other-mode --steps 500 --repeats 4 2> gpurun_out/tray_$1.err | grep '^{"metric"' > gpurun_out/tray_$1.json
  python - gpurun_out/tray_$1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "%.3f M" % (d["value"]/1e6), "err %.1e" % d["max_qpos_err"], "kernel_ms %.2f" % d["roofline"]["kernel_ms"], "handed %.4f" % (d.get("frac_envs_handed_over_to_the_full_kernel_in_the_last_launch") or 0))
PY
}
for rep in 1 2; do
  one w1_$rep 0 ""
  one w2_lds_$rep 1 ""
  [ -f $VL ] && one w2_regs_$rep 1 $VL
done
