# runs the exact-pd and drive-pd bench legs with every library under lib/variants (and the product library first)
mkdir -p gpurun_out
for so in cassie-mujoco-sim_amd/lib/libcassiemujoco.so cassie-mujoco-sim_amd/lib/variants/*.so; do
  [ -f "$so" ] || continue
  for mode in ${MODES:-exact-pd}; do
    v=$(CASSIE_LIB=$PWD/$so timeout 300 python bench.py --mode $mode --steps ${STEPS:-400} --warmup 50 --no-cpu-baseline --no-step-pd --no-other-mode --parity-envs 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f M  kernel %.3f ms  err %.1e' % (d['value']/1e6, d['roofline']['kernel_ms'], d['max_qpos_err']))")
    echo "$(basename $so) $mode: $v" | tee -a gpurun_out/variants.txt
  done
done
