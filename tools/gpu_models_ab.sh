# A/B of library variants on all three configs (drive-pd): product first, then every library under lib/variants
mkdir -p gpurun_out; rm -f gpurun_out/models_ab.txt
for rep in 1 2; do
for so in cassie-mujoco-sim_amd/lib/libcassiemujoco.so cassie-mujoco-sim_amd/lib/variants/*.so; do
  [ -f "$so" ] || continue
  for model in cassie cassie_hfield cassie_tray_box; do
    v=$(CASSIE_LIB=$PWD/$so timeout 300 python bench.py --model $model --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M  kernel %.3f ms  err %.1e' % (d['value']/1e6, d['roofline']['kernel_ms'], d['max_qpos_err']))")
    echo "$(basename $so) $model: $v" | tee -a gpurun_out/models_ab.txt
  done
done
done
